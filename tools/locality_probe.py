"""How much does cell ORDER cost?  Times the fused layers on the metric graph in three cell orders: the generator's (qhull
insertion order), Morton order of the cell centroids (what a locality permutation inside the plan could approach) and a random
permutation (what a CGAL-ordered real scene looks like: data/Ignatius has neighbour ids tens of thousands of rows apart)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from dgnn_amd import ops
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.graph import GraphPlan
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.synthetic import delaunay_tet_graph

dev = "cuda:0"
adj, cent, _ = delaunay_tet_graph(150000, 0)
n = adj.shape[0] // 4


def morton(c):
    q = np.clip(((c - c.min(0)) / (c.max(0) - c.min(0) + 1e-9) * 1023).astype(np.int64), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


rand = np.random.default_rng(0).permutation(n)


def graph_orders(base):
    """orders a plan could derive from the ADJACENCY alone (no coordinates), starting from the labelling `base` (a random one: the CGAL-ordered real
    scene): reverse Cuthill-McKee, and plain breadth-first order from cell 0 -- as new -> old index vectors like the others"""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import breadth_first_order, reverse_cuthill_mckee
    new_id = np.empty(n, np.int64)
    new_id[base] = np.arange(n)
    src = new_id[adj[:, 0]]
    dst = new_id[adj[:, 1]]
    A = sp.csr_matrix((np.ones(src.shape[0], np.int8), (dst, src)), shape=(n, n))
    rcm = reverse_cuthill_mckee(A, symmetric_mode=True)
    bfs = breadth_first_order(A, 0, directed=False, return_predecessors=False)
    if bfs.shape[0] < n:      # unreachable cells (none on this graph) go last
        bfs = np.concatenate([bfs, np.setdiff1d(np.arange(n), bfs)])
    return base[rcm], base[bfs]


orders = {"generator": np.arange(n), "morton": np.argsort(morton(cent), kind="stable"), "random": rand}
if os.environ.get("GRAPH_ORDERS", "1") == "1":
    orders["random->rcm"], orders["random->bfs"] = graph_orders(rand)
g = torch.Generator().manual_seed(0)
x = torch.randn(n, 29, generator=g)
ea = torch.randn(4 * n, 20, generator=g)
for dtype in ("f32", "bf16"):
    net = SurfaceNet(reconbench_pretrained(device=dev))
    net.load_state_dict(bench.load_weights())
    net = net.to(dev).eval()
    if dtype == "bf16":
        net.set_storage_dtype(torch.bfloat16)
    for name, order in orders.items():
        new_id = np.empty(n, np.int64)
        new_id[order] = np.arange(n)                      # old -> new
        dst = adj[:, 1].reshape(n, 4)[order]              # rows of new cell k = rows of old cell order[k]
        adj2 = np.stack([np.repeat(np.arange(n), 4), new_id[dst].reshape(-1)], 1)
        erow = (order[:, None] * 4 + np.arange(4)[None]).reshape(-1)
        data = Config(x=x[order].to(dev), edge_attr=ea[erow].to(dev), edge_index=torch.from_numpy(adj2.T.astype(np.int64)).to(dev))
        plan = GraphPlan(data.edge_index, n, n, hint=ops.PLAN_HINT_REFERENCE)
        h = net._storage_input(net._input_rows(data.x))
        ts = []
        for i in range(4):
            fn = lambda h=h, i=i: net._eval_layers(h, n, data.edge_attr, [plan] * 4, True, only=i)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
            h = fn()
        print("%-5s %-12s layers ms: %s  sum %.3f" % (dtype, name, " ".join("%.3f" % t for t in ts), sum(ts)), flush=True)
