"""oracle/extractor.py -- TEST INFRASTRUCTURE (never imported by reart_amd).

CPU restatement of the reference's correspondence extractor, networks/feature_extractor.py:10-49 (PointNet2Msg2.forward)
over networks/pointnet2_utils.py:194-348 (set abstraction with multi-scale grouping, group-all abstraction, feature
propagation), in eval mode (BatchNorm = the affine map of its running statistics):

  sampling / grouping     oracle.fps, oracle.ball_query           (C; pinned by pointnet_ops.npz)
  1x1 conv + BN + ReLU    one float32 matrix product per layer     y = relu(x (W s)^T + ((b - mean) s + beta)),  s = gamma / sqrt(var + eps)
  max over the group      numpy
  3-NN interpolation      oracle.three_interpolate                (C; pinned by three_interp.npz)

PINNED by tests/golden/extractor.npz: the reference's own module (seeded weights) on a 1024-point nao cloud -- sampled
coordinates bit-equal, features to float32 round-off of the layer products (tests/test_oracle_golden_cpu.py).  Weights come
as the module's state dict (numpy arrays under the reference's parameter names).
"""
import numpy as np

import oracle


def _fold(sd, conv, bn, eps=1e-5):
    """-> (Wt [Cin, Cout] float32, bias [Cout] float32) of conv followed by eval-mode batch norm."""
    w = np.asarray(sd[conv + ".weight"], np.float32)
    w = w.reshape(w.shape[0], -1)
    b = np.asarray(sd[conv + ".bias"], np.float32)
    s = np.asarray(sd[bn + ".weight"], np.float32) / np.sqrt(np.asarray(sd[bn + ".running_var"], np.float32) + np.float32(eps))
    return np.ascontiguousarray((w * s[:, None]).T), (b - np.asarray(sd[bn + ".running_mean"], np.float32)) * s + np.asarray(sd[bn + ".bias"], np.float32)


def _stack(x, sd, names):
    for conv, bn in names:
        Wt, b = _fold(sd, conv, bn)
        x = np.maximum(x @ Wt + b, np.float32(0))
    return x


def _index(points, idx):
    """points [B,N,C], idx [B,...] -> [B,...,C]  (index_points, networks/pointnet2_utils.py:58-71)"""
    B = points.shape[0]
    return points[np.arange(B).reshape((B,) + (1,) * (idx.ndim - 1)), idx]


def sa_msg(sd, prefix, xyz, feats, npoint, radii, nsamples, n_layers, start, cuda_mode):
    """PointNetSetAbstractionMsg.forward (:257-295), channel-last: xyz [B,N,3], feats [B,N,D] -> new_xyz [B,S,3], [B,S,sum C]"""
    fps = oracle.fps(xyz, npoint, start=start, cuda_mode=cuda_mode)
    new_xyz = _index(xyz, fps)
    outs = []
    for i, (radius, K) in enumerate(zip(radii, nsamples)):
        idx = oracle.ball_query(radius, K, xyz, new_xyz, cuda_mode=cuda_mode)                   # [B,S,K]
        g_xyz = _index(xyz, idx) - new_xyz[:, :, None, :]
        g = np.concatenate([_index(feats, idx), g_xyz], axis=-1) if feats is not None else g_xyz  # features first (:277-281)
        h = _stack(g.astype(np.float32), sd, [(f"{prefix}.conv_blocks.{i}.{j}", f"{prefix}.bn_blocks.{i}.{j}") for j in range(n_layers)])
        outs.append(h.max(axis=2))
    return new_xyz, np.concatenate(outs, axis=-1)


def sa_all(sd, prefix, xyz, feats, n_layers):
    """PointNetSetAbstraction with group_all (:209-235, 174-191): [xyz | feats] of every point, max over all points"""
    g = np.concatenate([xyz, feats], axis=-1).astype(np.float32)
    h = _stack(g, sd, [(f"{prefix}.mlp_convs.{j}", f"{prefix}.mlp_bns.{j}") for j in range(n_layers)])
    return h.max(axis=1, keepdims=True)                                                          # [B,1,C]


def fp(sd, prefix, xyz1, xyz2, points1, points2, n_layers):
    """PointNetFeaturePropagation.forward (:309-348)"""
    if xyz2.shape[1] == 1:
        interp = np.repeat(points2, xyz1.shape[1], axis=1)
    else:
        interp = oracle.three_interpolate(xyz1, xyz2, points2)
    x = np.concatenate([points1, interp], axis=-1).astype(np.float32) if points1 is not None else interp
    return _stack(x, sd, [(f"{prefix}.mlp_convs.{j}", f"{prefix}.mlp_bns.{j}") for j in range(n_layers)])


def forward(sd, xyz, fps_start=(None, None), cuda_mode=False, intermediates=False):
    """PointNet2Msg2.forward (networks/feature_extractor.py:31-49): xyz [B,3,N] -> [B,64,N].  ``fps_start`` = the FPS start
    indices of the two sampled levels (the reference's CPU path draws them with torch.randint; its CUDA path starts at 0)."""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    pts = np.ascontiguousarray(np.transpose(np.asarray(xyz, np.float32), (0, 2, 1)))             # [B,N,3]
    l1_xyz, l1 = sa_msg(sd, "sa1", pts, pts, 512, [0.05, 0.1, 0.2], [32, 64, 128], 3, fps_start[0], cuda_mode)
    l2_xyz, l2 = sa_msg(sd, "sa2", l1_xyz, l1, 128, [0.2, 0.4], [64, 128], 3, fps_start[1], cuda_mode)
    l3 = sa_all(sd, "sa3", l2_xyz, l2, 3)
    l2n = fp(sd, "fp3", l2_xyz, np.zeros_like(l2_xyz[:, :1]), l2, l3, 2)
    l1n = fp(sd, "fp2", l1_xyz, l2_xyz, l1, l2n, 2)
    l0n = fp(sd, "fp1", pts, l1_xyz, np.concatenate([pts, pts], axis=-1), l1n, 2)
    feat = _stack(l0n, sd, [("conv1", "bn1")])
    out = np.ascontiguousarray(np.transpose(feat, (0, 2, 1)))
    if intermediates:
        return out, dict(l1_xyz=l1_xyz, l1_points=l1, l2_xyz=l2_xyz, l2_points=l2)
    return out
