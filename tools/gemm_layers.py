#!/usr/bin/env python3
"""Per-layer time and useful TFLOP/s of the extractor's conv stacks (plain-input layers; B = 38 clouds of 4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reart_amd.networks.feature_extractor import mlp_layer

dev = torch.device("cuda:0")
B = 38
layers = []   # (name, rows, cin, cout, pool)
for K, ws in ((32, (32, 32, 64)), (64, (64, 64, 128)), (128, (64, 96, 128))):
    cin = 6
    for li, w in enumerate(ws):
        layers.append((f"sa1 K={K} {cin}->{w}", B * 512 * K, cin, w, K if li == 2 else 0)); cin = w
for K, ws in ((64, (128, 128, 256)), (128, (128, 196, 256))):
    cin = 323
    for li, w in enumerate(ws):
        layers.append((f"sa2 K={K} {cin}->{w}", B * 128 * K, cin, w, K if li == 2 else 0)); cin = w
cin = 515
for li, w in enumerate((256, 512, 1024)):
    layers.append((f"sa3 {cin}->{w}", B * 128, cin, w, 128 if li == 2 else 0)); cin = w
for name, rows, chain in (("fp3", B * 128, (1536, 256, 256)), ("fp2", B * 512, (576, 256, 128)), ("fp1", B * 4096, (134, 128, 128)), ("conv1", B * 4096, (128, 64))):
    for a, b in zip(chain[:-1], chain[1:]):
        layers.append((f"{name} {a}->{b}", rows, a, b, 0))
tot_t = tot_f = 0.0
for name, rows, cin, cout, pool in layers:
    X = torch.randn(rows, cin, device=dev)
    W = torch.randn(cin, cout, device=dev)
    b = torch.randn(cout, device=dev)
    for _ in range(2):
        mlp_layer(X, W, b, pool_k=pool)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        mlp_layer(X, W, b, pool_k=pool)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * rows * cin * cout
    tot_t += ms; tot_f += fl
    print(f"{name:22s} rows {rows:8d}  {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s")
print(f"total {tot_t:.3f} ms, {tot_f / tot_t / 1e9:.1f} TFLOP/s useful")
