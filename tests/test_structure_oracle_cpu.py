"""oracle/structure.py (numpy restatement of the reference's end-of-run structure extraction)
against tests/golden/structure.npz, which the reference's own functions produced (make_golden_structure.py) from
the shipped base-2 checkpoint.  Discrete results (labels, FPS indices, closest pairs, tree edges, orderings) are
bit-exact; floating-point results within 2e-6 absolute (fp32, the reference ran torch CPU kernels)."""
import os

import numpy as np
import pytest

import oracle
from oracle import structure as S

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "structure.npz"))
TOL = 2e-6


def close(a, b, tol=TOL):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=0, atol=tol)


def test_shipped_artefacts_agree_with_the_generated_golden():
    """The reference's shipped result (CUDA run) has the same segmentation, poses and undirected tree; only the
    order / direction in which its MST listed the edges differs (CUDA FPS vs CPU FPS tie order)."""
    assert G["shipped_same"][0] and G["shipped_same"][2]
    und = lambda e: sorted(tuple(sorted(x)) for x in e.tolist())
    assert und(G["new_conn"]) == und(G["shipped_conn"])
    kin = lambda c, p: sorted(zip(c.tolist(), p.tolist()))
    assert kin(G["edge_child"], G["edge_parent"]) == kin(G["kin_edge_child"], G["kin_edge_parent"])
    assert G["reverse_topo"][0] == G["kin_reverse_topo"][0] == G["root_part"]


def test_denoise_fps_pair_costs_and_first_merge_pass():
    cano, trans = G["cano"], G["trans0"]
    dn = S.denoise_seg_label(G["seg0"], cano, 20)
    np.testing.assert_array_equal(dn, G["seg_denoised"])
    idx = S.part_fps(cano, dn, G["uni0"], 20)
    np.testing.assert_array_equal(idx, G["fps_idx0"])
    pred = oracle.compute_pc_transform(cano, trans, dn)
    cd, pair, joint = S.part_pair_cost(cano, pred, idx)
    np.testing.assert_array_equal(cd, G["cano_dist0"])
    np.testing.assert_array_equal(pair, G["pair_idx0"])
    close(joint, G["joint0"])
    cost = (cd + joint + np.float32(1e4) * np.eye(len(cd), dtype=np.float32)).astype(np.float32)
    cand = S.mst(cost, G["uni0"])
    np.testing.assert_array_equal(cand, G["mst_merge0"])
    seg1, conn1 = S.merge_graph(dn, cand, trans, float(G["merge_thr"]))
    np.testing.assert_array_equal(seg1, G["seg_merge1"])
    np.testing.assert_array_equal(conn1, G["conn_merge1"])


def test_relative_screw_parameters_and_geo_cost():
    P = 20
    lab = np.arange(P)
    f = S.screw_fit(S.relative_trans(G["trans0"], np.repeat(lab, P), np.tile(lab, P)))
    off = ~np.eye(P, dtype=bool)           # the diagonal is inv(T) T: rounding noise decides its (unused) axis
    for k, gk in (("axis", "rel_axis"), ("moment", "rel_moment"), ("theta", "rel_theta"), ("distance", "rel_distance")):
        a = f[k].reshape((9, P, P) + f[k].shape[2:])
        close(a[:, off], G[gk][:, off])
    geo = S.geo_cost(G["trans0"], G["uni_merged"])
    close(geo, G["geo_cost"])


def test_wrappers_relabel_and_graph():
    cano, trans = G["cano"], G["trans0"]
    seg = S.merging_wrapper(G["seg_denoised"], trans, cano, float(G["merge_thr"]), int(G["merge_it"]))
    np.testing.assert_array_equal(seg, G["seg_merged"])
    conn = S.mst_wrapper(seg, trans, cano, 20, float(G["cano_dist_thr"]), float(G["lambda_joint"]))
    np.testing.assert_array_equal(conn, G["joint_connection_raw"])
    ns, nt, nc = S.extract_kinematic(seg, trans, conn)
    np.testing.assert_array_equal(ns, G["new_seg"])
    np.testing.assert_array_equal(nt, G["new_trans"])
    np.testing.assert_array_equal(nc, G["new_conn"])
    bg = S.build_graph(nc, nt)
    assert bg["root"] == int(G["root_part"])
    np.testing.assert_array_equal([e[0] for e in bg["edges"]], G["edge_child"])
    np.testing.assert_array_equal([e[1] for e in bg["edges"]], G["edge_parent"])
    np.testing.assert_array_equal(bg["reverse_topo"], G["reverse_topo"])
    np.testing.assert_array_equal(bg["nodes"], G["graph_nodes"])
    np.testing.assert_array_equal(np.concatenate([bg["paths_to_base"][p] for p in range(nt.shape[1])]), G["path_flat"])
    close(bg["axis"], G["axis_list"])
    close(bg["moment"], G["moment_list"])
    close(bg["theta"], G["theta_list"])


def test_energies():
    nt, nc, ns, cano = G["new_trans"], G["new_conn"], G["new_seg"], G["cano"]
    close(S.root_cost(nt), G["root_cost"])
    sc, rec = S.screw_cost(nt, nc)
    close(sc, G["screw_err"], 1e-8)
    close(rec, G["screw_recon"])
    pred = oracle.compute_pc_transform(cano, nt, ns)
    np.testing.assert_array_equal(pred, G["pred"])
    comp = np.concatenate([pred[:2], cano[None], pred[2:]])
    close(S.group_temporal_err(comp, ns), G["group_err"], 1e-8)
    close(S.ass_err(pred[:1], G["pc_list"][:1]), G["ass_err_frame0"], 1e-9)      # one 4096 x 4096 assignment


def test_screw_edge_cases():
    """identity, pure translation, rotation by pi, sub-threshold rotation, axis flipped towards (1,1,1), large rotation.
    For the no-rotation rows the reference's moment is 0.5 * (t x l + l x (t x l) / tan(5e-7)): rounding noise of t x l
    amplified 2e6 times -- compared only where there is a rotation."""
    l, m, th, d = S.transform_to_screw(G["screw_edge_T"])
    close(l, G["screw_edge_l"])
    close(th, G["screw_edge_theta"])
    close(d, G["screw_edge_d"])
    rot = np.abs(G["screw_edge_theta"]) > 1e-5
    assert rot.tolist() == [False, False, False, False, True, True, True]
    close(m[rot], G["screw_edge_m"][rot])
