"""End of an optimisation instance: structure extraction and the model-selection energy
(reference run_robot.py:224-330, in the order it calls things), on the GPU.

``extract_structure``  run_robot.py:231-243  denoise the segmentation, merge near-rigid neighbours, spanning tree,
                                            renumber the surviving parts
``energy_terms``       run_robot.py:306-321  assignment error (x100), screw error, group error, their sum -- the
                                            quantity the reference selects the canonical frame by (README.md:60)
``snapshot_metrics``   run_robot.py:245-266  flow EPE / accuracies / angle, Rand index, Chamfer and reconstruction
                                            errors against ground truth, when the sample has it
``kinematic_init``     run_robot.py:117-124  what KinematicModel is constructed from after a base run
"""
import torch

from .knn_cuda import KNN
from .utils import graph_utils as gu
from .utils import kinematic_utils as ku
from .utils.eval_utils import compute_chamfer_list, eval_flow, eval_seg
from .utils.model_utils import compute_ass_err, compute_group_temporal_err, compute_pc_transform


def extract_structure(seg_part, trans_list, cano_pc, knn=None, merge_thr=3e-2, merge_it=2, cano_dist_thr=1e-2,
                      lambda_joint=100, min_num=20, merge=True):
    """seg_part [N] (arg-max labels), trans_list [T-1,P,4,4], cano_pc [N,3] ->
    (seg_part [N] in 0..P'-1, trans_list [T-1,P',4,4], joint_connection [P'-1,2])."""
    knn = knn if knn is not None else KNN(k=1, transpose_mode=True)
    trans_list = trans_list.detach()
    seg_part = gu.denoise_seg_label(seg_part.clone(), cano_pc, knn, min_num=min_num)
    if merge and len(torch.unique(seg_part)) > 1:
        seg_part = gu.merging_wrapper(seg_part, trans_list, cano_pc, None, merge_thr, n_it=merge_it)
    joint_connection = gu.mst_wrapper(seg_part, trans_list, cano_pc, None, verbose=False, num_fps=20,
                                      cano_dist_thr=cano_dist_thr, joint_cost_weight=lambda_joint)
    return ku.extract_kinematic(seg_part, trans_list, joint_connection)


def energy_terms(cano_pc, pc_list, seg_part, trans_list, joint_connection, cano_idx, pred_pc_list=None):
    """-> dict(ass_err, screw_err, group_err, total_err) as floats (run_robot.py:306-314)."""
    if pred_pc_list is None:
        pred_pc_list = compute_pc_transform(cano_pc, trans_list, seg_part)
    ass_err = 100 * float(compute_ass_err(pred_pc_list, pc_list))
    screw_err = float(gu.compute_screw_cost(trans_list, joint_connection)) if joint_connection.numel() else 0.0
    complete = torch.cat((pred_pc_list[:cano_idx], cano_pc[None], pred_pc_list[cano_idx:]), dim=0)
    group_err = float(compute_group_temporal_err(complete, seg_part))
    return dict(ass_err=ass_err, screw_err=screw_err, group_err=group_err, total_err=ass_err + screw_err + group_err,
                lap_fallbacks=getattr(compute_ass_err, "last_fallbacks", 0))     # assignment problems solved on the host: 0 = none


def snapshot_values(cano_pc, seg_part, trans_list, cano_idx, sample):
    """The ground-truth metrics of a snapshot as ONE device tensor: -> (names, [len(names)] float64 tensor), no host sync and no
    data-dependent shape (a loop that prints hundreds of snapshots replays this from a captured graph: SnapshotPrinter)."""
    dev = cano_pc.device
    pred = compute_pc_transform(cano_pc, trans_list, seg_part)
    complete = torch.cat((pred[:cano_idx], cano_pc[None], pred[cano_idx:]), dim=0)
    vals, names = [], []
    if "gt_flow_list" in sample:
        gt = torch.as_tensor(sample["gt_flow_list"]).float().to(dev)
        vals += list(eval_flow(complete[1:] - complete[:-1], gt, acc1_thre=0.005, acc2_thre=0.01, as_tensors=True))
        names += ["epe", "acc5", "acc10", "angle"]
    if "gt_cano_part" in sample:
        gt_part = sample["gt_cano_part"]
        # the table holds every label of both sides: the ground truth's largest (read on the host when the sample lives there --
        # the loader's numpy arrays; a device tensor is taken to stay below 128) and the model's parts
        gt_max = int(gt_part.max()) if not (torch.is_tensor(gt_part) and gt_part.is_cuda) else 127
        vals.append(eval_seg(torch.as_tensor(gt_part).long().to(dev), seg_part, as_tensor=True,
                             num_labels=max(gt_max + 1, int(trans_list.shape[1]), 1)))
        names.append("ri")
    if "complete_gt_pc_list" in sample:
        gt = torch.as_tensor(sample["complete_gt_pc_list"]).float().to(dev)
        vals.append(((complete - gt) ** 2).sum(-1).sqrt().mean(1).mean())
        names.append("recon_err")
    return names, (torch.stack([v.double().reshape(()) for v in vals]) if vals else None)


def snapshot_scaled(names, host_values):
    """run_robot.py:245-266: centimetre scaling of the end-point and reconstruction errors."""
    return {k: 100 * v if k in ("epe", "recon_err") else v for k, v in zip(names, host_values)}


def snapshot_metrics(cano_pc, pc_list, seg_part, trans_list, cano_idx, sample=None, chamfer=True):
    """-> dict with cd_err (``chamfer``: the value is printed at the end of a run only, run_robot.py:316) and, when ``sample``
    carries the ground truth, epe / acc5 / acc10 / angle / ri / recon_err (run_robot.py:245-266; centimetre scaling as there).
    All of it is computed on the device and read with ONE copy (a run prints hundreds of snapshots)."""
    out = {}
    if chamfer:
        pred = compute_pc_transform(cano_pc, trans_list, seg_part)
        out["cd_err"] = 100 * compute_chamfer_list(pred, pc_list, reduction="mean")
    if sample is None:
        return out
    names, vals = snapshot_values(cano_pc, seg_part, trans_list, cano_idx, sample)
    if vals is not None:
        out.update(snapshot_scaled(names, vals.cpu().tolist()))
    return out


def kinematic_init(seg_part, trans_list, joint_connection):
    """-> kwargs for ``KinematicModel(pose_len, seg_part, cano_pc, knn, **kwargs)`` (run_robot.py:117-124)."""
    new_seg, new_trans, new_conn = ku.extract_kinematic(seg_part, trans_list, joint_connection)
    tree, root, axis, moment, theta, edge_index = ku.build_graph(new_conn, new_trans, verbose=False)
    return new_seg, dict(edge_index=edge_index, paths_to_base=tree.paths_to_base, reverse_topo=tree.reverse_topo,
                         axis_list=axis, moment_list=moment, theta_list=theta)


def finish_instance(model, cano_pc, pc_list, cano_idx, sample=None, **thresholds):
    """The whole tail for one optimised BaseModel: latest weights -> structure -> energies (+ metrics).
    -> dict(seg_part, trans_list, joint_connection, ass_err, screw_err, group_err, total_err, cd_err, ...)."""
    with torch.no_grad():
        _, seg_part, trans_list = model(cano_pc)
    seg_part, trans_list, conn = extract_structure(seg_part, trans_list, cano_pc, **thresholds)
    out = dict(seg_part=seg_part, trans_list=trans_list, joint_connection=conn)
    out.update(energy_terms(cano_pc, pc_list, seg_part, trans_list, conn, cano_idx))
    out.update(snapshot_metrics(cano_pc, pc_list, seg_part, trans_list, cano_idx, sample))
    return out
