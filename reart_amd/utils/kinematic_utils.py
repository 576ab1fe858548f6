"""Host-side mirror of ``fk`` from the reference's ``utils/kinematic_utils.py:151-198``."""
import numpy as np
import torch

from .. import _lib


def tree_arrays(edge_index, reverse_topo):
    """Flatten the reference's joint-tree dicts (edge_index: "child_parent" -> edge id,
    reverse_topo: parts from root to leaf; networks/model.py:79-93) into int32 arrays
    (parent, edge_of_part, order)."""
    P = len(reverse_topo)
    assert sorted(int(v) for v in reverse_topo) == list(range(P))
    parent = np.full(P, -1, np.int32)
    edge_of = np.full(P, -1, np.int32)
    for key, e in edge_index.items():
        c, p = (int(v) for v in key.split("_"))
        parent[c], edge_of[c] = p, int(e)
    order = np.asarray([int(v) for v in reverse_topo], np.int32)
    assert (parent < 0).sum() == 1, "the joint graph must be a tree with one root"
    return parent, edge_of, order


class _FK(torch.autograd.Function):
    """pc_trans, trans_list = rigid_apply(fk(...)); gradients to axis / moment / theta / distance."""

    @staticmethod
    def forward(ctx, x, part, axis, moment, theta, distance, parent, edge_of, order):
        _lib.require_gpu(x, part, axis, moment, theta)
        x, part = x.contiguous().float(), part.contiguous().long()
        axis, moment, theta = axis.contiguous().float(), moment.contiguous().float(), theta.contiguous().float()
        dist = None if distance is None else distance.contiguous().float()
        B, E = theta.shape
        P, N = parent.shape[0], x.shape[0]
        trans = torch.empty((B, P, 4, 4), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        rc = L.reart_fk_forward(_lib.ptr(parent), _lib.ptr(edge_of), _lib.ptr(order), P, _lib.ptr(axis),
                                _lib.ptr(moment), _lib.ptr(theta), _lib.ptr(dist), B, E, _lib.ptr(trans), _lib.stream())
        _lib.check(rc, "reart_fk_forward")
        out = torch.empty((B, N, 3), dtype=torch.float32, device=x.device)
        rc = L.reart_compute_pc_transform(_lib.ptr(x), _lib.ptr(trans), _lib.ptr(part), N, P, B, _lib.ptr(out),
                                          _lib.stream())
        _lib.check(rc, "reart_compute_pc_transform")
        ctx.save_for_backward(x, part, axis, moment, theta, trans, parent, edge_of, order)
        ctx.dist = dist
        ctx.mark_non_differentiable(trans)
        return out, trans

    @staticmethod
    def backward(ctx, g_out, g_trans):
        x, part, axis, moment, theta, trans, parent, edge_of, order = ctx.saved_tensors
        dist = ctx.dist
        B, E = theta.shape
        P, N = parent.shape[0], x.shape[0]
        G = g_out.contiguous().float()
        g_axis, g_moment, g_theta = torch.empty_like(axis), torch.empty_like(moment), torch.empty_like(theta)
        g_dist = None if dist is None else torch.empty_like(dist)
        L = _lib.lib()
        ws = _lib.workspace(L.reart_fk_backward_workspace_bytes(P, B, E), x.device)
        rc = L.reart_fk_backward(_lib.ptr(x), _lib.ptr(part), _lib.ptr(G), N, _lib.ptr(parent), _lib.ptr(edge_of),
                                 _lib.ptr(order), P, _lib.ptr(axis), _lib.ptr(moment), _lib.ptr(theta), _lib.ptr(dist),
                                 B, E, _lib.ptr(trans), _lib.ptr(g_axis), _lib.ptr(g_moment), _lib.ptr(g_theta),
                                 _lib.ptr(g_dist), _lib.ptr(ws), ws.numel(), _lib.stream())
        _lib.check(rc, "reart_fk_backward")
        return None, None, g_axis, g_moment, g_theta, g_dist, None, None, None


def fk(paths_to_base, reverse_topo, edge_index, axis_list, moment_list, theta_list, distance_list=None,
       joint_type_list=None):
    """Forward kinematics over screw joints -> [T, P, 4, 4] (utils/kinematic_utils.py:151-198).
    ``paths_to_base`` is accepted for signature compatibility; with parts visited root to leaf the
    reference's path walk always stops at the first edge, so the tree's parent links suffice."""
    dev = theta_list.device
    parent, edge_of, order = (torch.from_numpy(a).to(dev) for a in tree_arrays(edge_index, reverse_topo))
    theta, dist = _effective_joint_values(theta_list, distance_list, joint_type_list)
    dummy_x = torch.zeros((1, 3), device=dev)
    dummy_part = torch.zeros((1,), dtype=torch.long, device=dev)
    _, trans = _FK.apply(dummy_x, dummy_part, axis_list, moment_list, theta, dist, parent, edge_of, order)
    return trans


def _effective_joint_values(theta_list, distance_list, joint_type_list):
    """utils/kinematic_utils.py:174-186: prismatic joints run with theta = 1e-6 and their distance,
    revolute joints with their theta and distance = 1e-6 (also the default without a type list)."""
    if joint_type_list is None:
        return theta_list, distance_list
    pris = torch.tensor([jt == "prismatic" for jt in joint_type_list], device=theta_list.device)
    theta = torch.where(pris[None, :], torch.full_like(theta_list, 1e-6), theta_list)
    dist = torch.where(pris[None, :], distance_list, torch.full_like(distance_list, 1e-6))
    return theta, dist


# ---------------------------------------------------------------------------------------------------------
# Structure extraction: relabelling and the joint tree (utils/kinematic_utils.py:18-149).  The graph has at most
# num_parts nodes: its bookkeeping is host integer work; the screw parameters of the edges are one kernel launch.
def extract_kinematic(seg_part, trans_list, joint_connection):
    """utils/kinematic_utils.py:18-34: renumber the surviving parts 0..P-1 in label order."""
    uni_label = torch.unique(seg_part, sorted=True)
    if joint_connection.numel():       # a single surviving part has no edges
        assert torch.equal(torch.unique(joint_connection, sorted=True), uni_label)
    new_seg = torch.searchsorted(uni_label, seg_part.contiguous())
    new_conn = torch.searchsorted(uni_label, joint_connection.contiguous())
    return new_seg, trans_list[:, uni_label], new_conn


def _paths_to(adj, root):
    """Breadth-first parent links towards ``root`` in an undirected tree -> {node: [node, ..., root]}."""
    parent, order = {root: None}, [root]
    for u in order:
        for v in adj[u]:
            if v not in parent:
                parent[v] = u
                order.append(v)
    paths = {}
    for v in parent:
        p, cur = [], v
        while cur is not None:
            p.append(cur)
            cur = parent[cur]
        paths[v] = p
    return paths


class JointTree:
    """The directed joint tree (edges child -> parent) with the orderings the reference derives from its networkx
    graph: ``edges`` in the order ``G.edges()`` lists them, ``nodes`` in insertion order, ``paths_to_base``
    (= ``nx.shortest_path(G, target=root)``) and ``reverse_topo`` (root to leaf)."""

    def __init__(self, edges_list, root):
        und, first_seen = {}, []
        for a, b in edges_list:
            for x, y in ((a, b), (b, a)):
                if x not in und:
                    und[x] = []
                    first_seen.append(x)
                und[x].append(y)
        paths = _paths_to(und, root)
        if len(paths) != len(first_seen):
            raise AssertionError("invalid tree structure")
        new_edges = []
        for part in first_seen:                           # utils/kinematic_utils.py:41-47
            path = paths[part]
            for i in range(len(path) - 1):
                if (path[i], path[i + 1]) not in new_edges:
                    new_edges.append((path[i], path[i + 1]))
        assert len(new_edges) == len(first_seen) - 1, "invalid tree structure"
        self.root = root
        self.nodes, self.parent = [], {}
        for c, p in new_edges:                            # DiGraph insertion order
            for x in (c, p):
                if x not in self.nodes:
                    self.nodes.append(x)
            self.parent[c] = p
        self.edges = [(c, self.parent[c]) for c in self.nodes if c in self.parent]
        self.paths_to_base = {}
        for v in self.nodes:
            p, cur = [v], v
            while cur in self.parent:
                cur = self.parent[cur]
                p.append(cur)
            self.paths_to_base[v] = p
        # topological order by generations (children before parents), reversed: root first
        indeg = {v: 0 for v in self.nodes}
        for c, p in self.edges:
            indeg[p] += 1
        gen, topo = [v for v in self.nodes if indeg[v] == 0], []
        while gen:
            nxt = []
            for u in gen:
                topo.append(u)
                if u in self.parent:
                    indeg[self.parent[u]] -= 1
                    if indeg[self.parent[u]] == 0:
                        nxt.append(self.parent[u])
            gen = nxt
        self.reverse_topo = list(reversed(topo))

    def number_of_nodes(self):
        return len(self.nodes)


def to_DAG(edges_list, root_node):
    """utils/kinematic_utils.py:37-53 on a plain edge list -> JointTree (children point to parents)."""
    return JointTree([(int(a), int(b)) for a, b in edges_list], int(root_node))


def build_graph(edges_list, trans_list, verbose=False, root_part=None, revolute_only=True, return_joint_type=False):
    """utils/kinematic_utils.py:57-139: root = the part that moves least; per edge (child, parent) the mean screw
    axis / moment of inv(T_parent) T_child and its angle per frame.
    -> (tree, root_part, axis_list [E,3], moment_list [E,3], theta_list [T,E], edge_index) and, with
    ``revolute_only=False``, distance_list [T,E] (+ joint types): the cheaper of the revolute / prismatic fits."""
    from .graph_utils import compute_root_cost, screw_fit

    edges = torch.as_tensor(edges_list).cpu().tolist()
    P = trans_list.shape[1]
    assert sorted({int(x) for e in edges for x in e}) == list(range(P))
    if root_part is None:
        root_part = int(compute_root_cost(trans_list).argmin().item())
    if verbose:
        print("root part id", root_part)
    G = to_DAG(edges, root_part)
    pairs = torch.tensor([[p, c] for c, p in G.edges], dtype=torch.int32, device=trans_list.device)
    # build_graph averages over all frames per edge (its compute_mean_screw_param calls see E = 1)
    f = screw_fit(trans_list, pairs, plain_mean=True, want=("screw",))
    screw = f["screw"]
    axis_list, moment_list = f["mean"][:, 0:3].contiguous(), f["mean"][:, 3:6].contiguous()
    theta, distance = screw[..., 6], screw[..., 7]
    edge_index = {"_".join([str(c), str(p)]): k for k, (c, p) in enumerate(G.edges)}
    if revolute_only:
        no_rot = (theta.abs() < 1e-6) | ((theta - np.pi).abs() < 1e-6)
        assert int(no_rot.sum()) == 0
        joint_type_list = ["revolute"] * len(G.edges)
        print("joint types at each edge: {}".format(joint_type_list))
        return G, root_part, axis_list, moment_list, theta.contiguous(), edge_index
    per_edge = screw_fit_per_edge_costs(trans_list, pairs)
    pris = per_edge[:, 1] <= per_edge[:, 0]
    joint_type_list = ["prismatic" if bool(b) else "revolute" for b in pris]
    tiny = torch.full_like(theta, 1e-6)
    theta_list = torch.where(pris[None, :], tiny, theta)
    distance_list = torch.where(pris[None, :], distance, tiny)
    print("joint types at each edge: {}".format(joint_type_list))
    if return_joint_type:
        return G, root_part, axis_list, moment_list, theta_list, distance_list, edge_index, joint_type_list
    return G, root_part, axis_list, moment_list, theta_list, distance_list, edge_index


def screw_fit_per_edge_costs(trans_list, pairs):
    """(revolute, prismatic) cost of every edge fitted ON ITS OWN (utils/kinematic_utils.py:100-126: there the
    rotation residual ``F.mse_loss`` is averaged over one edge, not over all edges as in compute_geo_cost)."""
    from .graph_utils import screw_fit

    rows = [screw_fit(trans_list, pairs[e:e + 1], plain_mean=True)["cost"][0, :2] for e in range(pairs.shape[0])]
    return torch.stack(rows)


def edge_index2edges(edge_index):
    """utils/kinematic_utils.py:142-149."""
    return [[int(v) for v in name.split("_")] for name in edge_index.keys()]


def ik(dataset, model, device, verbose=True, vis=True, save_dir=None, **ikargs):
    """Retargeting error (utils/kinematic_utils.py:200-266): for every novel pose of the sequence fit the model's
    free motion parameters (one ``theta`` row for a KinematicModel, one 6D / translation proposal per part for a
    BaseModel) to ONE point per ground-truth part with Adam(amsgrad, lr 0.1, 200 iterations), then measure how far
    the whole canonical cloud lands from its ground-truth novel position (mean Euclidean distance, x100).
    ``dataset``: an object with ``[0]`` -> sample, ``pose_list``, ``cano_idx``, ``novel_pose_list`` (the mirror
    ``reart_amd.dataset.Sequence`` or the reference's).  ``vis`` / ``save_dir`` are accepted; nothing is drawn."""
    from ..networks.model import BaseModel
    from .dataset_utils import sparse_sample_novel_state

    sample = dataset[0]
    cano_pose = dataset.pose_list[dataset.cano_idx]
    cano_pc = torch.from_numpy(sample["cano_pc"]).to(device)
    errs = []
    for novel_pose in dataset.novel_pose_list:
        novel = sparse_sample_novel_state(sample["cano_pc"], sample["gt_cano_part"], cano_pose, novel_pose, 1)
        errs.append(ik_single(model, cano_pc, novel, device, verbose=verbose, **ikargs)[0])
    return np.array(errs).mean()


def ik_single(model, cano_pc, novel_sample, device, n_iter=200, verbose=False, **ikargs):
    """One novel pose of ``ik`` -> (retarget error x100, fitted kwargs)."""
    from ..networks.model import BaseModel

    src = torch.from_numpy(np.asarray(novel_sample["sparse_cano_pc"])).float().to(device)
    tgt = torch.from_numpy(np.asarray(novel_sample["sparse_novel_pc"])).float().to(device)
    kwargs = {}
    if isinstance(model, BaseModel):
        kwargs["tau"] = ikargs.get("tau", 1.0)
        kwargs["proposal_6d"] = torch.tensor([[[1.0, 0, 0, 0, 1, 0]]], device=device).repeat((1, model.num_parts, 1)).requires_grad_(True)
        kwargs["proposal_t"] = torch.zeros((1, model.num_parts, 3), device=device).requires_grad_(True)
        opt_list = [kwargs["proposal_6d"], kwargs["proposal_t"]]
    else:
        kwargs["theta_list"] = (1e-6 * torch.ones((1, model.axis_list.shape[0]), device=device)).requires_grad_(True)
        opt_list = [kwargs["theta_list"]]
    optimizer = torch.optim.Adam(opt_list, lr=1e-1, amsgrad=True)
    for _ in range(n_iter):
        pc_trans, _, _ = model(src, **kwargs)
        loss = ((pc_trans.squeeze(0) - tgt) ** 2).sum()
        if verbose:
            print(f"Loss: {loss:.3f}")
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
    with torch.no_grad():
        pc_trans, _, _ = model(cano_pc.float(), **kwargs)
    gt = torch.from_numpy(np.asarray(novel_sample["novel_pc"])).float().to(device)
    err = 100 * float((pc_trans.squeeze(0) - gt).pow(2).sum(-1).sqrt().mean())
    if verbose:
        print(f"Novel retarget err: {err:.3f}")
    return err, kwargs
