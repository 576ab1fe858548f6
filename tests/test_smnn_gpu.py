"""GPU: SMNN descriptor matching vs a plain PyTorch restatement of the reference's algorithm
(cdist -> topk(2) -> ratio test both ways -> mutual), and the run_robot-style driver end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_smnn(d1, d2, th=0.9):
    dm = torch.cdist(d1.double(), d2.double())
    v, i2 = torch.topk(dm, 2, dim=1, largest=False)
    ok1 = (v[:, 0] / v[:, 1]) <= th
    v2, i1 = torch.topk(dm.t(), 2, dim=1, largest=False)
    ok2 = (v2[:, 0] / v2[:, 1]) <= th
    pairs = []
    for i in range(d1.shape[0]):
        j = int(i2[i, 0])
        if ok1[i] and ok2[j] and int(i1[j, 0]) == i:
            pairs.append((i, j))
    margin = torch.minimum((v[:, 0] / v[:, 1] - th).abs().min(), (v2[:, 0] / v2[:, 1] - th).abs().min())
    return np.asarray(pairs, np.int64).reshape(-1, 2), float(margin)


def test_match_smnn_vs_torch(dev):
    from reart_amd.utils.flow_utils import match_smnn

    g = torch.Generator().manual_seed(0)
    base = torch.randn((900, 64), generator=g)
    d1 = base + 0.15 * torch.randn((900, 64), generator=g)
    d2 = torch.cat([base[torch.randperm(900, generator=g)[:700]] + 0.15 * torch.randn((700, 64), generator=g),
                    torch.randn((200, 64), generator=g)])
    ref, margin = _ref_smnn(d1, d2)
    assert margin > 1e-5 and len(ref) > 100  # no borderline ratio in this draw: the sets must be identical
    _, got = match_smnn(d1.to(dev), d2.to(dev))
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_match_smnn_vs_reference_golden(dev, tag):
    """tests/golden/smnn.npz: the reference's own match_smnn (utils/flow_utils.py:48-100) on these descriptors."""
    import os

    from reart_amd.utils.flow_utils import match_smnn

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "smnn.npz"))
    assert float(g[f"margin_{tag}"]) > 1e-4      # no borderline ratio: the match set is well defined in fp32
    _, got = match_smnn(torch.from_numpy(g[f"d1_{tag}"]).to(dev), torch.from_numpy(g[f"d2_{tag}"]).to(dev))
    np.testing.assert_array_equal(got.cpu().numpy(), g[f"idx_{tag}"])


def test_run_robot_driver_synthetic(dev, tmp_path):
    """The drop-in loop: fused phase, hand-over to the assignment-loss phase, checkpoint keys."""
    from reart_amd.run_robot import build_parser, main

    args = build_parser().parse_args(["--synthetic", "--synthetic_frames", "5", "--num_points", "512", "--cano_idx", "2",
                                      "--n_iter", "60", "--assign_iter", "40", "--use_assign_loss", "--use_flow_loss",
                                      "--snapshot_gap", "20", "--downsample", "8", "--save_root", str(tmp_path)])
    model = main(args)
    ck = torch.load(next(tmp_path.rglob("model.pth.tar")), weights_only=False)
    assert set(ck) == {"state_dict", "tau", "cano_idx"} and ck["cano_idx"] == 2
    assert {"proposal_6d", "proposal_t", "seg_head.model.0.weight", "seg_head.model.2.weight"} <= set(ck["state_dict"])
    assert all(torch.isfinite(p).all() for p in model.parameters())
    txt = next(tmp_path.rglob("result.txt")).read_text()
    assert "assign_refreshes: 4" in txt and "lap_fallbacks: 0" in txt        # refreshes at 40, 45, 50, 55; none solved on the host


def test_run_robot_end_of_run_files_and_kinematic_from_base_result(dev, tmp_path):
    """End of run (run_robot.py:224-356): result.pkl / result.txt / model.pth.tar with the reference's keys, then the
    kinematic model built from that base result (run_robot.py:101-124) optimises and saves its tree."""
    import pickle

    from reart_amd.run_robot import build_parser, main

    base = ["--synthetic", "--synthetic_frames", "6", "--num_points", "1024", "--cano_idx", "2", "--snapshot_gap", "500"]
    main(build_parser().parse_args(base + ["--n_iter", "1500", "--save_root", str(tmp_path / "base")]))
    res_path = next((tmp_path / "base").rglob("result.pkl"))
    with open(res_path, "rb") as f:
        res = pickle.load(f)
    assert {"pred_cano_part", "pred_pose_list", "cano_idx", "joint_connection", "cano_pc", "pc_list"} <= set(res)
    P = res["pred_pose_list"].shape[1]
    assert res["pred_pose_list"].shape == (5, P, 4, 4) and res["pred_cano_part"].max() == P - 1
    assert len(res["joint_connection"]) == P - 1
    txt = next((tmp_path / "base").rglob("result.txt")).read_text()
    assert "total_err" in txt and "ass_err" in txt and "cd_err" in txt
    if P > 1:
        # BASELINE.json config 5: kinematic projection with the assignment loss on, downsample 2
        model = main(build_parser().parse_args(base + ["--model", "kinematic", "--base_result_path", str(res_path),
                                                       "--use_assign_loss", "--assign_iter", "10", "--downsample", "2",
                                                       "--n_iter", "40", "--save_root", str(tmp_path / "kin")]))
        ck = torch.load(next((tmp_path / "kin").rglob("model.pth.tar")), weights_only=False)
        assert {"state_dict", "tau", "cano_idx", "seg_part", "cano_pc", "edge_index", "paths_to_base", "reverse_topo"} <= set(ck)
        assert len(ck["edge_index"]) == P - 1 and all(torch.isfinite(p).all() for p in model.parameters())


def test_run_robot_on_a_sequence_directory_with_ground_truth_and_retargeting(dev, tmp_path):
    """--seq_path in the reference's on-disk layout (written by reart_amd.synthetic.export_sequence): the mirror loader,
    ground-truth metrics in result.txt, and for the kinematic model the retargeting error to the novel poses (ik)."""
    from reart_amd.run_robot import build_parser, main
    from reart_amd.synthetic import export_sequence

    seq_dir = str(tmp_path / "toy")
    export_sequence(seq_dir, T=6, n_parts=4, pts_per_part=256, seed=2, n_novel=2)
    base = ["--seq_path", seq_dir, "--num_points", "1024", "--cano_idx", "2", "--snapshot_gap", "1000"]
    main(build_parser().parse_args(base + ["--n_iter", "2000", "--save_root", str(tmp_path / "base")]))
    txt = next((tmp_path / "base").rglob("result.txt")).read_text()
    for key in ("recon_err", "epe", "acc5", "ri", "cd_err", "total_err", "retarget_err: 9999"):
        assert key in txt, (key, txt)
    res_path = next((tmp_path / "base").rglob("result.pkl"))
    import pickle

    with open(res_path, "rb") as f:
        res = pickle.load(f)
    assert {"gt_flow_list", "gt_pose_list", "complete_gt_pc_list", "pred_cano_part", "joint_connection"} <= set(res)
    if res["pred_pose_list"].shape[1] > 1:
        main(build_parser().parse_args(base + ["--model", "kinematic", "--base_result_path", str(res_path), "--n_iter", "60",
                                               "--save_root", str(tmp_path / "kin")]))
        txt = next((tmp_path / "kin").rglob("result.txt")).read_text()
        err = float([l for l in txt.splitlines() if l.startswith("retarget_err")][0].split(":")[1])
        assert 0.0 <= err < 100.0


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mnn_matching_vs_reference_golden(dev, tag):
    """matching="mnn" of compute_corr_list_filter (utils/flow_utils.py:126-137) against the reference's own function
    (tests/golden/mnn.npz: its k = 1 KNN both ways + find_mutual_correspondences on the smnn.npz descriptors)."""
    import os

    from reart_amd.utils.flow_utils import compute_corr_list_filter, find_mutual_correspondences

    G = os.path.join(os.path.dirname(__file__), "golden")
    g, m = np.load(os.path.join(G, "smnn.npz")), np.load(os.path.join(G, "mnn.npz"))
    n = int(m[f"n_{tag}"])
    assert float(m[f"gap_{tag}"]) > 1e-4        # no near-tie between a best and second-best descriptor: well defined in fp32
    frames = torch.stack([torch.from_numpy(g[f"d1_{tag}"][:n]), torch.from_numpy(g[f"d2_{tag}"][:n])]).to(dev)
    extractor = lambda x: frames.transpose(1, 2).contiguous()            # [T, 64, N] like PointNet2Msg2
    src, tgt = compute_corr_list_filter(torch.zeros(2, n, 3, device=dev), extractor, None, matching="mnn")
    assert len(src) == 1
    np.testing.assert_array_equal(src[0].cpu().numpy(), m[f"src_{tag}"])
    np.testing.assert_array_equal(tgt[0].cpu().numpy(), m[f"tgt_{tag}"])
    # the helper of the same name
    nn01 = torch.cdist(frames[0], frames[1]).argmin(1)
    nn10 = torch.cdist(frames[1], frames[0]).argmin(1)
    s2, t2 = find_mutual_correspondences(nn01, nn10)
    np.testing.assert_array_equal(s2.cpu().numpy(), m[f"src_{tag}"])
    np.testing.assert_array_equal(t2.cpu().numpy(), m[f"tgt_{tag}"])
    with pytest.raises(ValueError):
        compute_corr_list_filter(torch.zeros(2, n, 3, device=dev), extractor, None, matching="other")
