"""Host-side mirror of the reference's ``networks/feature_extractor.py`` (PointNet2Msg2, the
64-d per-point correspondence descriptor net, ``:10-49``) over the HIP kernels: FPS and ball
query (``reart_fps`` / ``reart_ball_query``), gather-fused 1x1-conv stacks with max-pool on the
fp32 matrix cores (``reart_mlp_layer``) and 3-NN interpolation (``reart_three_interpolate``).

Module / parameter names equal the reference's (``sa1.conv_blocks.0.0.weight``, ``sa1.bn_blocks...``,
``sa3.mlp_convs.0``, ``fp1.mlp_bns.1``, ``conv1``, ``bn1``) so ``corr_model.pth.tar``-style checkpoints
load with ``strict=True`` (feature_extractor.py:62-86).  Inference only (the reference freezes the
extractor, ``rec_freeze`` :52-59): BatchNorm uses running statistics and is folded into the conv.
"""
import torch
import torch.nn as nn

from .. import _lib
from .pointnet2_utils import farthest_point_sample, index_points, query_ball_point


def _fold(conv, bn):
    """eval-mode BN folded into the 1x1 conv -> (Wt [Cin, Cout] contiguous, bias [Cout]).  The folded pair is kept on the
    conv module until one of the six tensors it was made from changes (in-place updates and ``load_state_dict`` bump
    ``_version``; a re-assigned tensor is another object): nine tiny launches per layer and forward otherwise,
    a sixth of the extractor's device time.  The entry holds references to its source tensors, so an address cannot be
    recycled under it.  NOT seen: writes through ``.data`` (``p.data.copy_()``, ``p.data.mul_()`` -- they bypass the
    version counter); after such an update call ``PointNet2Msg2.invalidate_fold()`` (``.train()``, ``.to()`` / ``.cuda()``
    / ``.float()`` and ``load_state_dict`` on the extractor do it themselves)."""
    src = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((id(t), t.data_ptr(), t._version) if t is not None else None for t in src) + (bn.eps,)
    hit = getattr(conv, "_reart_folded", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    Wt, bf = _fold_now(conv, bn)
    conv._reart_folded = (key, Wt, bf, src)          # src: keeps the keyed tensors (and their addresses) alive
    return Wt, bf


def _fold_now(conv, bn):
    w = conv.weight.detach().reshape(conv.weight.shape[0], -1).float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    wf = w * scale[:, None]
    bf = (b - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
    return wf.t().contiguous(), bf.contiguous()


def mlp_layer(X, Wt, bias, relu=True, pool_k=0, out=None, out_col=0, gather=None):
    """Y = relu(X Wt + bias) [max over pool_k consecutive rows] via reart_mlp_layer.
    gather = dict(idx [B,S,K] i64, F [B*Npts,D] or None, Q [B*Npts,3], C [B*S,3] or None, Npts, xyz_first)."""
    L = _lib.lib()
    Cin, Cout = Wt.shape
    if gather is None:
        rows, ldx = X.shape
        args = (_lib.ptr(X), ldx, None, 0, 0, 0, None, 0, None, None, 0)
    else:
        B, S, K = gather["idx"].shape
        rows = B * S * K
        F = gather.get("F")
        args = (None, 0, _lib.ptr(gather["idx"]), K, S, gather["Npts"], _lib.ptr(F), 0 if F is None else F.shape[1],
                _lib.ptr(gather["Q"]), _lib.ptr(gather.get("C")), int(gather.get("xyz_first", 0)))
    orows = rows // pool_k if pool_k else rows
    if out is None:
        out = torch.empty((orows, Cout), dtype=torch.float32, device=Wt.device)
    rc = L.reart_mlp_layer(*args, _lib.ptr(Wt), _lib.ptr(bias), rows, Cin, Cout, int(relu), pool_k, _lib.ptr(out),
                           out.shape[1], out_col, _lib.stream())
    _lib.check(rc, "reart_mlp_layer")
    return out


OVERLAP_SAMPLING = True      # the second level's FPS / ball queries on a side stream under the first level's GEMMs
_SIDE = {}


def _side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


CHAIN3 = {(32, 32, 64, 32), (64, 64, 128, 64), (64, 96, 128, 128)}    # (C1, C2, C3, nsample) built into reart_mlp_chain3
CHAIN3_WIDE = {(128, 128, 256, 64), (128, 196, 256, 128)}             # sa2's scales, built into reart_mlp_chain3_wide
FUSE_CHAIN = True      # False: every layer its own launch (the same bits; tests compare the two)


def mlp_chain3(folded, gather, out, out_col):
    """Three gathered 1x1-conv layers + max over the group in ONE launch (reart_mlp_chain3): ``folded`` = [(Wt, bias)] x 3,
    ``gather`` as for ``mlp_layer`` with 3 feature columns; the result lands in out[:, out_col:out_col + C3]."""
    L = _lib.lib()
    (W1, b1), (W2, b2), (W3, b3) = folded
    B, S, K = gather["idx"].shape
    rc = L.reart_mlp_chain3(_lib.ptr(gather["idx"]), K, S, gather["Npts"], _lib.ptr(gather["F"]), _lib.ptr(gather["Q"]),
                            _lib.ptr(gather["C"]), _lib.ptr(W1), _lib.ptr(b1), W1.shape[1], _lib.ptr(W2), _lib.ptr(b2), W2.shape[1],
                            _lib.ptr(W3), _lib.ptr(b3), W3.shape[1], B * S * K, _lib.ptr(out), out.shape[1], out_col, _lib.stream())
    _lib.check(rc, "reart_mlp_chain3")
    return out


def mlp_chain3_wide(folded, gather, out, out_col):
    """The same for a scale with D (% 4 == 0) feature columns and wide layers (reart_mlp_chain3_wide: the weights stream
    through LDS, a workgroup carries 128 rows through the three layers)."""
    L = _lib.lib()
    (W1, b1), (W2, b2), (W3, b3) = folded
    B, S, K = gather["idx"].shape
    F = gather["F"]
    ws = _lib.workspace(L.reart_mlp_chain3_wide_workspace_bytes(F.shape[1], W1.shape[1], W2.shape[1], W3.shape[1]), F.device)
    rc = L.reart_mlp_chain3_wide(_lib.ptr(gather["idx"]), K, S, gather["Npts"], _lib.ptr(F), F.shape[1], _lib.ptr(gather["Q"]),
                                 _lib.ptr(gather["C"]), _lib.ptr(W1), _lib.ptr(b1), W1.shape[1], _lib.ptr(W2), _lib.ptr(b2),
                                 W2.shape[1], _lib.ptr(W3), _lib.ptr(b3), W3.shape[1], B * S * K, _lib.ptr(out), out.shape[1], out_col,
                                 _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "reart_mlp_chain3_wide")
    return out


def three_interpolate(xyz1, xyz2, points2, out, out_col):
    """PointNetFeaturePropagation's 3-NN interpolation written into out[:, out_col:out_col+D]."""
    L = _lib.lib()
    B, N, _ = xyz1.shape
    S2, D = points2.shape[1], points2.shape[2]
    ws = _lib.workspace(L.reart_three_interpolate_workspace_bytes(B, N, S2), xyz1.device)
    rc = L.reart_three_interpolate(_lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(points2), B, N, S2, D, _lib.ptr(out),
                                   out.shape[1], out_col, _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, "reart_three_interpolate")


class _SAMsg(nn.Module):
    """Parameter container + forward of PointNetSetAbstractionMsg (networks/pointnet2_utils.py:238-295)."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list = npoint, radius_list, nsample_list
        self.conv_blocks, self.bn_blocks = nn.ModuleList(), nn.ModuleList()
        for mlp in mlp_list:
            convs, bns, last = nn.ModuleList(), nn.ModuleList(), in_channel + 3
            for out in mlp:
                convs.append(nn.Conv2d(last, out, 1))
                bns.append(nn.BatchNorm2d(out))
                last = out
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)
        self.out_channels = sum(m[-1] for m in mlp_list)

    def sample(self, xyz, start=None, cuda_mode=None):
        """The part of the level that depends on COORDINATES only: farthest point sampling and the ball queries of every
        scale -> (new_xyz [B,S,3], [idx [B,S,K] per scale]).  The caller may run it ahead of the previous level's
        feature stacks (another stream)."""
        fps = farthest_point_sample(xyz, self.npoint, start=start, cuda_mode=cuda_mode)
        new_xyz = index_points(xyz, fps).contiguous()
        return new_xyz, [query_ball_point(radius, K, xyz, new_xyz, cuda_mode=cuda_mode)
                         for radius, K in zip(self.radius_list, self.nsample_list)]

    def run(self, xyz, feats, start=None, cuda_mode=None, sampled=None):
        """xyz [B,N,3], feats [B,N,D] (channel-last) -> new_xyz [B,S,3], new_feats [B,S,sum C]."""
        B, N, _ = xyz.shape
        S = self.npoint
        new_xyz, idx_list = sampled if sampled is not None else self.sample(xyz, start=start, cuda_mode=cuda_mode)
        out = torch.empty((B * S, self.out_channels), dtype=torch.float32, device=xyz.device)
        col = 0
        Q = xyz.reshape(B * N, 3)
        F = feats.reshape(B * N, -1).contiguous()
        for i, (radius, K) in enumerate(zip(self.radius_list, self.nsample_list)):
            idx = idx_list[i]
            h = None
            n_layers = len(self.conv_blocks[i])
            widths = tuple(c.weight.shape[0] for c in self.conv_blocks[i])
            if FUSE_CHAIN and n_layers == 3 and F.shape[1] == 3 and widths + (K,) in CHAIN3:
                # sa1: the activations between the three layers stay in LDS (one launch per scale instead of three)
                folded = [_fold(conv, bn) for conv, bn in zip(self.conv_blocks[i], self.bn_blocks[i])]
                mlp_chain3(folded, dict(idx=idx, F=F, Q=Q, C=new_xyz.reshape(B * S, 3), Npts=N), out, col)
                col += widths[-1]
                continue
            if (FUSE_CHAIN and n_layers == 3 and widths + (K,) in CHAIN3_WIDE and F.shape[1] % 4 == 0 and (B * S * K) % 128 == 0
                    and F.data_ptr() % 16 == 0):
                # sa2: the same with the weights streamed through LDS
                folded = [_fold(conv, bn) for conv, bn in zip(self.conv_blocks[i], self.bn_blocks[i])]
                mlp_chain3_wide(folded, dict(idx=idx, F=F, Q=Q, C=new_xyz.reshape(B * S, 3), Npts=N), out, col)
                col += widths[-1]
                continue
            for j, (conv, bn) in enumerate(zip(self.conv_blocks[i], self.bn_blocks[i])):
                Wt, b = _fold(conv, bn)
                last = j == n_layers - 1
                if j == 0:  # grouped [features | xyz - centre], features first (:277-281)
                    h = mlp_layer(None, Wt, b, gather=dict(idx=idx, F=F, Q=Q, C=new_xyz.reshape(B * S, 3), Npts=N,
                                                           xyz_first=0),
                                  pool_k=K if last else 0, out=out if last else None, out_col=col if last else 0)
                else:
                    h = mlp_layer(h, Wt, b, pool_k=K if last else 0, out=out if last else None,
                                  out_col=col if last else 0)
            col += self.conv_blocks[i][-1].weight.shape[0]
        return new_xyz, out.reshape(B, S, self.out_channels)


class _SAAll(nn.Module):
    """PointNetSetAbstraction with group_all=True (networks/pointnet2_utils.py:194-235, 174-191)."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for out in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out))
            last = out

    def run(self, xyz, feats):
        B, N, _ = xyz.shape
        if N not in (32, 64, 128):
            raise NotImplementedError("group_all pooling is built for 32/64/128 points (the extractor has 128)")
        idx = torch.arange(N, device=xyz.device, dtype=torch.int64).expand(B, 1, N).contiguous()
        h, n_layers = None, len(self.mlp_convs)
        for j, (conv, bn) in enumerate(zip(self.mlp_convs, self.mlp_bns)):
            Wt, b = _fold(conv, bn)
            last = j == n_layers - 1
            if j == 0:  # [xyz | features], xyz first (sample_and_group_all :186-188)
                h = mlp_layer(None, Wt, b, gather=dict(idx=idx, F=feats.reshape(B * N, -1).contiguous(),
                                                       Q=xyz.reshape(B * N, 3), C=None, Npts=N, xyz_first=1),
                              pool_k=N if last else 0)
            else:
                h = mlp_layer(h, Wt, b, pool_k=N if last else 0)
        return h  # [B, C]


class _FP(nn.Module):
    """PointNetFeaturePropagation (networks/pointnet2_utils.py:298-348)."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for out in mlp:
            self.mlp_convs.append(nn.Conv1d(last, out, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out))
            last = out

    def run(self, xyz1, xyz2, points1, points2):
        """xyz1 [B,N,3], xyz2 [B,S,3], points1 [B,N,D1] or None, points2 [B,S,D2] -> [B,N,C]."""
        B, N, _ = xyz1.shape
        S, D2 = points2.shape[1], points2.shape[2]
        D1 = 0 if points1 is None else points1.shape[2]
        X = torch.empty((B * N, D1 + D2), dtype=torch.float32, device=xyz1.device)
        if D1:
            X[:, :D1] = points1.reshape(B * N, D1)
        if S == 1:
            X[:, D1:] = points2.expand(B, N, D2).reshape(B * N, D2)
        else:
            three_interpolate(xyz1.contiguous(), xyz2.contiguous(), points2.contiguous(), X, D1)
        h = X
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            Wt, b = _fold(conv, bn)
            h = mlp_layer(h, Wt, b)
        return h.reshape(B, N, -1)


class PointNet2Msg2(nn.Module):
    """feature_extractor.py:10-49; forward(xyz [B,3,N]) -> [B,out_dim,N]."""

    def __init__(self, out_dim, normal_channel=False):
        super().__init__()
        if normal_channel:
            raise NotImplementedError("the reference builds the extractor with normal_channel=False (:63)")
        self.out_dim, self.normal_channel = out_dim, normal_channel
        self.sa1 = _SAMsg(512, [0.05, 0.1, 0.2], [32, 64, 128], 3, [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = _SAMsg(128, [0.2, 0.4], [64, 128], 128 + 128 + 64, [[128, 128, 256], [128, 196, 256]])
        self.sa3 = _SAAll(512 + 3, [256, 512, 1024])
        self.fp3 = _FP(1536, [256, 256])
        self.fp2 = _FP(576, [256, 128])
        self.fp1 = _FP(134, [128, 128])
        self.conv1 = nn.Conv1d(128, out_dim, 1)
        self.bn1 = nn.BatchNorm1d(out_dim)

    def invalidate_fold(self):
        """Drop every cached BatchNorm fold (see ``_fold``): needed by hand only after writes through ``.data``."""
        for m in self.modules():
            if hasattr(m, "_reart_folded"):
                del m._reart_folded
        return self

    def train(self, mode=True):
        self.invalidate_fold()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):         # .to() / .cuda() / .float() / .half()
        self.invalidate_fold()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_fold()
        return super().load_state_dict(*args, **kwargs)

    @torch.no_grad()
    def forward(self, xyz, fps_start=None, cuda_mode=None):
        """``cuda_mode`` None follows ``pointnet2_utils.CUDA`` (True: the rules of a GPU run of the reference);
        ``cuda_mode=False`` selects the CPU-fallback rules, for which ``fps_start`` = (start1 [B], start2 [B])
        injects the FPS start indices the reference draws from torch's RNG."""
        if self.training:
            raise RuntimeError("PointNet2Msg2 is inference-only here (call .eval()); the reference freezes it")
        _lib.require_gpu(xyz)
        B, _, N = xyz.shape
        pts = xyz.permute(0, 2, 1).contiguous().float()  # [B,N,3]
        s1, s2 = fps_start if fps_start is not None else (None, None)
        samp1 = self.sa1.sample(pts, start=s1, cuda_mode=cuda_mode)
        samp2 = None
        if OVERLAP_SAMPLING:
            # the second level's sampling (FPS 512 -> 128: a latency chain on B workgroups; its ball queries) needs the first
            # level's coordinates only: it runs on a side stream under the first level's feature stacks
            main, side = torch.cuda.current_stream(xyz.device), _side_stream(xyz.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                samp2 = self.sa2.sample(samp1[0], start=s2, cuda_mode=cuda_mode)
                for t_ in (samp2[0], *samp2[1]):
                    t_.record_stream(main)           # allocated on the side stream, consumed on the main one
        l1_xyz, l1 = self.sa1.run(pts, pts, sampled=samp1)                           # [B,512,3], [B,512,320]
        if samp2 is not None:
            main.wait_stream(side)
        l2_xyz, l2 = self.sa2.run(l1_xyz, l1, start=s2, cuda_mode=cuda_mode, sampled=samp2)   # [B,128,3], [B,128,512]
        l3 = self.sa3.run(l2_xyz, l2)                                                # [B,1024]
        l2n = self.fp3.run(l2_xyz, l2_xyz[:, :1], l2, l3[:, None, :])                # [B,128,256]
        l1n = self.fp2.run(l1_xyz, l2_xyz, l1, l2n)                                  # [B,512,128]
        l0n = self.fp1.run(pts, l1_xyz, torch.cat([pts, pts], dim=2), l1n)           # [B,N,128]
        Wt, b = _fold(self.conv1, self.bn1)
        feat = mlp_layer(l0n.reshape(B * N, -1), Wt, b)                              # [B*N,64]
        return feat.reshape(B, N, self.out_dim).permute(0, 2, 1).contiguous()


def get_extractor(args):
    """feature_extractor.py:62-86 without the DataParallel wrapper: checkpoints saved from the
    wrapped model carry a ``module.`` prefix, which is stripped here."""
    import warnings

    model = PointNet2Msg2(out_dim=64)
    ckpt = torch.load(args.corr_model_path, map_location="cpu")
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    sd = {k.replace("net.", "").replace("module.", "", 1): v for k, v in sd.items()}
    missing = set(model.state_dict().keys()) - set(sd.keys())
    if missing:
        warnings.warn("Missing keys ! : {}".format(missing))
    model.load_state_dict(sd, strict=True)
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    return model.cuda()
