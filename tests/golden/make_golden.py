#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (read-only
checkout at /root/reference) in this container.  Run once here; the .npz files are committed,
the reference itself never travels to the GPU box.

    python tests/golden/make_golden.py

Stand-ins (ours, installed in sys.modules before the reference is imported; nothing in the
reference tree is modified or copied):
  * chamferdist._C  -> oracle.knn_points / knn_points_backward   (UNPINNED third-party contract)
  * knn_cuda.KNN    -> oracle.knn_cuda (Euclidean distances)      (UNPINNED third-party contract)
  * imageio, apted, apted.helpers, trimesh -> empty modules (only needed to satisfy imports of
    utils/viz_utils.py:9, utils/ted_utils.py:5-6, dataset/dataset_real.py:3; never executed)
Everything else that ends up in the fixtures is computed by the reference's own Python.
"""
import os
import pickle
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REART_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import oracle  # noqa: E402  (test infrastructure)


# ------------------------------------------------------------------ stand-ins
def _install_standins():
    cd = types.ModuleType("chamferdist")
    c = types.ModuleType("chamferdist._C")

    def knn_points_idx(p1, p2, l1, l2, K, version):
        d, i = oracle.knn_points(p1.detach().numpy(), p2.detach().numpy(), l1.numpy(), l2.numpy(), K)
        return torch.from_numpy(i), torch.from_numpy(d)

    def knn_points_backward(p1, p2, l1, l2, idx, g):
        g1, g2 = oracle.knn_points_backward(p1.detach().numpy(), p2.detach().numpy(), idx.numpy(),
                                            g.detach().numpy(), l1.numpy(), l2.numpy())
        return torch.from_numpy(g1), torch.from_numpy(g2)

    c.knn_points_idx, c.knn_points_backward = knn_points_idx, knn_points_backward
    cd._C = c
    sys.modules["chamferdist"], sys.modules["chamferdist._C"] = cd, c

    kc = types.ModuleType("knn_cuda")

    class KNN(torch.nn.Module):
        def __init__(self, k, transpose_mode=False):
            super().__init__()
            self.k, self._t = k, transpose_mode

        def forward(self, ref, query):
            with torch.no_grad():
                if not self._t:
                    ref, query = ref.transpose(1, 2), query.transpose(1, 2)
                d, i = oracle.knn_cuda(ref.contiguous().numpy(), query.contiguous().numpy(), self.k, True)
                d, i = torch.from_numpy(d), torch.from_numpy(i)
                if not self._t:
                    d, i = d.transpose(1, 2).contiguous(), i.transpose(1, 2).contiguous()
                return d, i

    kc.KNN = KNN
    sys.modules["knn_cuda"] = kc
    for name in ("imageio", "trimesh"):
        sys.modules[name] = types.ModuleType(name)
    ap = types.ModuleType("apted")
    ap.APTED, ap.Config = object, object
    aph = types.ModuleType("apted.helpers")
    aph.Tree = object
    sys.modules["apted"], sys.modules["apted.helpers"] = ap, aph


_install_standins()

from utils.chamfer import ChamferDistance, knn_points  # noqa: E402
from networks.loss import recon_loss, flow_loss  # noqa: E402
from networks.model import BaseModel, KinematicModel  # noqa: E402
from utils.flow_utils import blend_anchor_motion  # noqa: E402
from utils.model_utils import compute_pc_transform, tau_cosine  # noqa: E402
from utils.eval_utils import compute_chamfer_list  # noqa: E402
from dataset.dataset_robot import Sequence  # noqa: E402
import screw_se3  # noqa: E402
from knn_cuda import KNN  # noqa: E402
from networks import pointnet2_utils as pn2  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", path, {k: np.asarray(v).shape for k, v in arrays.items()})


def gumbel_with_seed(seed, shape):
    """The noise F.gumbel_softmax draws when torch.manual_seed(seed) precedes the call."""
    torch.manual_seed(seed)
    return -torch.empty(shape).exponential_().log()


def main():
    torch.set_num_threads(8)
    sample = Sequence(os.path.join(REF, "demo_data/data/nao"), num_points=4096, cano_idx=2)[0]
    cano_full = torch.from_numpy(sample["cano_pc"]).float()
    pc_full = torch.from_numpy(sample["pc_list"]).float()
    sub = np.random.default_rng(2).permutation(4096)[:512]
    cano, pcs = cano_full[sub].contiguous(), pc_full[:, sub].contiguous()

    # ---- G1/G2: knn_points, ChamferDistance, recon_loss + grad (reference autograd path)
    rng = np.random.default_rng(12)
    a = torch.from_numpy(rng.uniform(-0.35, 0.35, (2, 512, 3)).astype(np.float32))
    b = torch.from_numpy(rng.uniform(-0.35, 0.35, (2, 512, 3)).astype(np.float32))
    kn = knn_points(a, b, K=1)
    k3 = knn_points(a, b, K=3)
    src = pcs[:3].clone().requires_grad_(True)
    tgt = pcs[3:6].clone()
    cd, fi, bi = ChamferDistance()(src, tgt, bidirectional=True, return_index=True)
    loss = recon_loss(src, tgt, ChamferDistance())
    loss.backward()
    # independent cross-check from the reference: KD-tree Chamfer (utils/eval_utils.py:39-66)
    kd = compute_chamfer_list(src.detach().numpy(), tgt.numpy(), reduction="sum")
    save("chamfer", a=a, b=b, k1_d=kn.dists, k1_i=kn.idx, k3_d=k3.dists, k3_i=k3.idx,
         src=src.detach(), tgt=tgt, cd=cd.detach(), fwd_idx=fi, bwd_idx=bi,
         recon_loss=loss.detach(), grad_src=src.grad, kdtree_sum=kd)

    # ---- G3/G4: flow_loss (+robust) and blend_anchor_motion
    gt = torch.from_numpy(rng.normal(0, 0.02, (4, 300, 3)).astype(np.float32))
    pred = (gt + torch.from_numpy(rng.normal(0, 0.01, (4, 300, 3)).astype(np.float32))).requires_grad_(True)
    pred.data[0, :5] *= 200.0  # exercise the Huber linear branch
    mask = torch.from_numpy(rng.uniform(size=(4, 300)) < 0.7)
    out = {}
    for robust in (False, True):
        pred.grad = None
        l = flow_loss(gt, pred, flow_mask_list=mask, robust=robust)
        l.backward()
        out["loss_r%d" % robust], out["grad_r%d" % robust] = l.detach(), pred.grad.clone()
    pred.grad = None
    l = flow_loss(gt, pred)  # default mask of ones
    l.backward()
    out["loss_nomask"], out["grad_nomask"] = l.detach(), pred.grad.clone()
    q = pcs[0]
    ref_idx = rng.permutation(512)[:300]
    ref = pcs[1][ref_idx].contiguous()
    ref_flow = torch.from_numpy(rng.normal(0, 0.03, (300, 3)).astype(np.float32))
    ref_flow[:40] *= 0.01  # tiny flows: makes the mask's first clause fail for some points
    bl, bm = blend_anchor_motion(q, ref, ref_flow, KNN(k=3, transpose_mode=True), return_mask=True)
    q_far = q * 4.0  # far queries: both mask clauses false for some points
    bl2, bm2 = blend_anchor_motion(q_far, ref, ref_flow, KNN(k=3, transpose_mode=True), return_mask=True)
    save("flow", gt=gt, pred=pred.detach(), mask=mask, query=q, ref=ref, ref_flow=ref_flow,
         blend=bl, blend_mask=bm, query_far=q_far, blend_far=bl2, blend_mask_far=bm2, **out)

    # ---- G5: BaseModel.forward / autograd backward from the shipped base-2 checkpoint
    ck = torch.load(os.path.join(REF, "demo_data/pretrained/nao/base-2/model.pth.tar"), map_location="cpu",
                    weights_only=False)
    model = BaseModel(num_parts=20, pose_len=9)
    model.load_state_dict(ck["state_dict"], strict=False)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g5 = {}
    for tag, tau, seed in (("a", 1.0, 5), ("b", 3.7, 6)):
        noise = gumbel_with_seed(seed, (512, 20))
        torch.manual_seed(seed)
        model.zero_grad()
        out_pc, seg, trans = model(cano, tau=tau)
        Gw = torch.from_numpy(rng.normal(size=tuple(out_pc.shape)).astype(np.float32))
        (out_pc * Gw).sum().backward()
        g5.update({f"noise_{tag}": noise, f"tau_{tag}": np.float32(tau), f"out_{tag}": out_pc.detach(),
                   f"seg_{tag}": seg, f"trans_{tag}": trans.detach(), f"G_{tag}": Gw,
                   f"g6d_{tag}": model.proposal_6d.grad.clone(), f"gt_{tag}": model.proposal_t.grad.clone(),
                   f"gW1_{tag}": model.seg_head.model[0].weight.grad[:, :, 0].clone(),
                   f"gb1_{tag}": model.seg_head.model[0].bias.grad.clone(),
                   f"gW2_{tag}": model.seg_head.model[2].weight.grad[:, :, 0].clone()})
    # random (non-orthonormal) 6D parameters: exercises the Gram-Schmidt backward properly
    with torch.no_grad():
        model.proposal_6d.add_(torch.from_numpy(rng.normal(0, 0.3, (9, 20, 6)).astype(np.float32)))
    noise = gumbel_with_seed(7, (512, 20))
    torch.manual_seed(7)
    model.zero_grad()
    out_pc, seg, trans = model(cano, tau=2.0)
    Gw = torch.from_numpy(rng.normal(size=tuple(out_pc.shape)).astype(np.float32))
    (out_pc * Gw).sum().backward()
    g5.update({"p6d_c": model.proposal_6d.detach().clone(), "noise_c": noise, "tau_c": np.float32(2.0),
               "out_c": out_pc.detach(), "seg_c": seg, "trans_c": trans.detach(), "G_c": Gw,
               "g6d_c": model.proposal_6d.grad.clone(), "gt_c": model.proposal_t.grad.clone()})
    save("base_model", cano=cano, W1=sd["seg_head.model.0.weight"][:, :, 0], b1=sd["seg_head.model.0.bias"],
         W2=sd["seg_head.model.2.weight"][:, :, 0], p6d=sd["proposal_6d"], pt=sd["proposal_t"], **g5)

    # ---- G6: 6D / SE(3) / screw maps on the 90 nao transforms + edge cases
    with open(os.path.join(REF, "demo_data/pretrained/nao/base-2/result_14999.pkl"), "rb") as f:
        res = pickle.load(f)
    poses = torch.from_numpy(res["pred_pose_list"]).float().reshape(-1, 4, 4)
    d6 = torch.cat([screw_se3.matrix_to_rotation_6d(poses[:, :3, :3]),
                    torch.from_numpy(rng.normal(size=(40, 6)).astype(np.float32))])
    Rm = screw_se3.rotation_6d_to_matrix(d6)
    logt = torch.from_numpy(rng.normal(0, 1.0, (64, 6)).astype(np.float32))
    logt[:8, 3:] *= 1e-3   # rotations below the eps clamp (|w|^2 < 1e-4)
    logt[8:12, 3:] = 0.0
    se3 = screw_se3.se3_exp_map(logt)
    l = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(48, 3)).astype(np.float32)), dim=-1)
    m = torch.from_numpy(rng.normal(0, 0.3, (48, 3)).astype(np.float32))
    th = torch.from_numpy(rng.uniform(-3.0, 3.0, 48).astype(np.float32))
    dd = torch.from_numpy(rng.normal(0, 0.1, 48).astype(np.float32))
    th[:4] = 1e-6; dd[:4] = 0.25          # the "prismatic" placeholder of kinematic_utils.py:176-186
    th[4:6] = 0.0                         # exact zero: no_rot branch
    th[6] = float(np.float32(np.pi))      # |theta - pi| < 1e-6: no_rot branch
    dd[8:12] = 1e-6                       # revolute default distance
    expc = screw_se3.screw_param_to_exponential_coordinates(l, m, th, dd)
    T = screw_se3.transform_from_exponential_coordinates(expc)
    inv = screw_se3.inverse_transformation(poses)
    save("se3", d6=d6, R=Rm, logt=logt, se3=se3, l=l, m=m, theta=th, d=dd, expc=expc, T=T,
         poses=poses, poses_inv=inv)

    # ---- G7: fk + KinematicModel.forward from the shipped kinematic-2 checkpoint
    ckk = torch.load(os.path.join(REF, "demo_data/pretrained/nao/kinematic-2/model.pth.tar"), map_location="cpu",
                     weights_only=False)
    kmodel = KinematicModel(pose_len=9, seg_part=ckk["seg_part"], cano_pc=ckk["cano_pc"],
                            knn=KNN(k=1, transpose_mode=True), edge_index=ckk["edge_index"],
                            paths_to_base=ckk["paths_to_base"], reverse_topo=ckk["reverse_topo"])
    kmodel.load_state_dict(ckk["state_dict"], strict=True)
    kin_in = cano_full[:1024].contiguous()
    kout, kseg, ktrans = kmodel(kin_in)
    Gk = torch.from_numpy(rng.normal(size=tuple(kout.shape)).astype(np.float32))
    kmodel.zero_grad()
    (kout * Gk).sum().backward()
    kfull, _, _ = kmodel(cano_full)
    cd_k = 100 * compute_chamfer_list(kfull.detach().numpy(), sample["pc_list"], reduction="mean").mean()
    P = kmodel.num_parts
    parent = np.full(P, -1, np.int32)
    edge_of = np.full(P, -1, np.int32)
    for key, e in ckk["edge_index"].items():
        c, p = (int(v) for v in key.split("_"))
        parent[c], edge_of[c] = p, e
    save("kinematic", cano_pc=ckk["cano_pc"].float(), seg_part=ckk["seg_part"].long(), input_pc=kin_in,
         axis=kmodel.axis_list.detach(), moment=kmodel.moment_list.detach(), theta=kmodel.theta_list.detach(),
         parent=parent, edge_of_part=edge_of, order=np.asarray(ckk["reverse_topo"], np.int32),
         out=kout.detach(), seg=kseg, trans=ktrans.detach(), G=Gk,
         g_axis=kmodel.axis_list.grad, g_moment=kmodel.moment_list.grad, g_theta=kmodel.theta_list.grad,
         cd_x100=np.float64(cd_k))

    # ---- G8: FPS (start index recorded) and ball query (CPU-fallback semantics)
    xyz = torch.stack([cano_full, pc_full[0]])
    # normalise like pc_normalize-scaled inputs of the extractor: radii 0.05..0.4 assume ~unit scale
    xyzn = (xyz - xyz.mean(dim=1, keepdim=True))
    xyzn = xyzn / xyzn.norm(dim=-1).max()
    torch.manual_seed(3)
    start = torch.randint(0, 4096, (2,), dtype=torch.long)
    torch.manual_seed(3)
    fps1 = pn2.farthest_point_sample(xyzn, 512)
    assert (fps1[:, 0] == start).all()
    new_xyz = pn2.index_points(xyzn, fps1)
    torch.manual_seed(4)
    start2 = torch.randint(0, 512, (2,), dtype=torch.long)
    torch.manual_seed(4)
    fps2 = pn2.farthest_point_sample(new_xyz, 128)
    new_xyz2 = pn2.index_points(new_xyz, fps2)
    bq = {}
    for r, K, src_, ctr, tag in ((0.05, 32, xyzn, new_xyz, "a"), (0.1, 64, xyzn, new_xyz, "b"),
                                 (0.2, 128, xyzn, new_xyz, "c"), (0.2, 64, new_xyz, new_xyz2, "d"),
                                 (0.4, 128, new_xyz, new_xyz2, "e")):
        bq["bq_" + tag] = pn2.query_ball_point(r, K, src_, ctr)
    save("pointnet_ops", xyz=xyzn, start1=start, fps1=fps1, start2=start2, fps2=fps2, **bq)

    # ---- G10: known answers of the committed artefacts (quality numbers)
    pose = torch.from_numpy(res["pred_pose_list"]).float()
    part = torch.from_numpy(res["pred_cano_part"]).long()
    pred = compute_pc_transform(cano_full, pose, part)
    cd_res = 100 * compute_chamfer_list(pred.numpy(), sample["pc_list"], reduction="mean").mean()
    model2 = BaseModel(num_parts=20, pose_len=9)
    model2.load_state_dict(ck["state_dict"], strict=False)
    torch.manual_seed(2)
    with torch.no_grad():
        o2, _, _ = model2(cano_full, tau=1.0)
    cd_ck = 100 * compute_chamfer_list(o2.numpy(), sample["pc_list"], reduction="mean").mean()
    save("known_answers", pose=pose, part=part, pred_sub=pred[:, sub], cano_sub=cano, sub=sub,
         cd_result_x100=np.float64(cd_res), cd_ckpt_x100=np.float64(cd_ck), cd_kin_x100=np.float64(cd_k))
    print("known answers: result %.6f  ckpt %.6f  kinematic %.6f" % (cd_res, cd_ck, cd_k))

    # ---- G11: 10-step relaxation trajectory on the 512-point nao subsample, injected noise
    torch.manual_seed(2)
    m3 = BaseModel(num_parts=20, pose_len=9)
    init = {k: v.detach().clone() for k, v in m3.state_dict().items()}
    seg_params = filter(lambda p: p.requires_grad, m3.seg_head.parameters())
    opt = torch.optim.Adam([{"params": [m3.proposal_6d, m3.proposal_t], "lr": 1e-2},
                            {"params": seg_params, "lr": 1e-3}], lr=1e-3, weight_decay=0)
    chamfer = ChamferDistance()
    n_iter = 10
    noises, losses, taus = [], [], []
    for i in range(n_iter):
        tau = tau_cosine(i + 1, 15000, 1, 5)
        noises.append(gumbel_with_seed(1000 + i, (512, 20)))
        torch.manual_seed(1000 + i)
        pc_trans, _, _ = m3(cano, tau=tau)
        loss = recon_loss(pc_trans, pcs, chamfer)
        losses.append(loss.item()); taus.append(tau)
        opt.zero_grad(); loss.backward(); opt.step()
    fin = {k: v.detach().clone() for k, v in m3.state_dict().items()}
    save("trajectory", cano=cano, pcs=pcs, noises=torch.stack(noises), losses=np.asarray(losses, np.float64),
         taus=np.asarray(taus, np.float64),
         W1_0=init["seg_head.model.0.weight"][:, :, 0], b1_0=init["seg_head.model.0.bias"],
         W2_0=init["seg_head.model.2.weight"][:, :, 0],
         W1_f=fin["seg_head.model.0.weight"][:, :, 0], b1_f=fin["seg_head.model.0.bias"],
         W2_f=fin["seg_head.model.2.weight"][:, :, 0], p6d_f=fin["proposal_6d"], pt_f=fin["proposal_t"])
    print("trajectory losses", losses)


if __name__ == "__main__":
    main()
