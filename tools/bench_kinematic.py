#!/usr/bin/env python3
"""BASELINE.json config 5 at the synthetic headline size: kinematic projection (model=kinematic, assignment loss on,
downsample 2 -> 19 matrices of 2048 x 2048 every assign_gap = 5 iterations) from a base result."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reart_amd.run_robot import build_parser, main

tmp = tempfile.mkdtemp()
common = ["--synthetic", "--synthetic_frames", "20", "--num_points", "4096", "--cano_idx", "10", "--snapshot_gap", "100000"]
main(build_parser().parse_args(common + ["--use_flow_loss", "--n_iter", "3000", "--save_root", tmp + "/base"]))
res = os.path.join(tmp, "base", "nao", "result.pkl")
n = int(os.environ.get("ITERS", 300))
torch.cuda.synchronize(); t0 = time.perf_counter()
main(build_parser().parse_args(common + ["--model", "kinematic", "--base_result_path", res, "--use_assign_loss", "--assign_iter", "0",
                                         "--downsample", "2", "--n_iter", str(n), "--save_root", tmp + "/kin"]))
torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"kinematic projection: {n} iterations in {el:.1f} s incl. model construction and end of run = {n / el:.1f} it/s")
