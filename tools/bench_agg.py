"""Aggregate forward / backward micro-benchmark on the blocks of one training batch (2048 targets, 4 hops, 1M-tet scene) at the
Static model's layer widths: average launch time over 20 replays (HIP events) and achieved GB/s on the compulsory bytes.
    DGNN_LIB=dgnn_amd/variants/x.so python tools/bench_agg.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgnn_amd._lib as L
if os.environ.get("DGNN_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["DGNN_LIB"])
import numpy as np, torch
from dgnn_amd import ops
from dgnn_amd.graph import plan_for
from dgnn_amd.sampler import NeighborSampler
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
dev = "cuda:0"
adj, _, _ = delaunay_tet_graph(150000, 0)
n = adj.shape[0] // 4
ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=dev)
idx = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:2048].to(dev)
_, n_id, adjs = NeighborSampler(ei, sizes=[-1] * 4, num_nodes=n, batch_size=2048).sample(idx)
def t(f, it=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
tot_f = tot_b = 0
for (e, e_id, size), c in zip(adjs, (28, 64, 128, 128)):
    plan = plan_for(e, size[0], size[1]); tp = plan.transposed
    E = e.size(1)
    x = torch.randn(size[0], c, device=dev); da = torch.randn(size[1], c, device=dev)
    We = torch.randn(c, 20, device=dev) * 0.2; be = torch.randn(c, device=dev) * 0.1
    rows = plan.edge_rows if plan.edge_rows is not None else e_id.to(torch.int32)
    t_rows = torch.index_select(rows, 0, tp[2])
    fwd = t(lambda: ops.aggregate_fwd(plan.rowptr, plan.src, rows, size[1], x, ea, We, be))
    bwd = t(lambda: ops.aggregate_bwd(tp[0], tp[1], t_rows, size[0], plan.rowptr, x, da, ea, We, be, need_dx=c != 28))
    bf = (size[0] * c * 4 + E * 88 + size[1] * c * 4 + size[1] * 4)
    bb = (size[0] * c * 4 * (2 if c != 28 else 1) + E * 88 + size[1] * c * 4 + size[0] * 4)
    print("n_src %6d n_dst %6d E %6d c_in %3d: fwd %6.1f us %5.0f GB/s   bwd %6.1f us %5.0f GB/s" % (size[0], size[1], E, c, fwd, bf / fwd / 1e3, bwd, bb / bwd / 1e3))
    tot_f += fwd; tot_b += bwd
print("sum over the 4 layers: fwd %.1f us  bwd %.1f us" % (tot_f, tot_b))
