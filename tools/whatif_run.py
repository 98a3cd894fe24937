"""Times layer 3 (128->128) with a what-if build of the library (results are garbage, only the time matters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgnn_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from dgnn_amd import ops
from dgnn_amd.graph import GraphPlan
from dgnn_amd.synthetic import delaunay_tet_graph
dev = "cuda:0"
adj, _, _ = delaunay_tet_graph(150000, 0)
n = adj.shape[0] // 4
plan = GraphPlan(torch.from_numpy(adj.T.astype(np.int64)).to(dev), n, n)
g = torch.Generator(device=dev).manual_seed(0)
for c_in, c_out in ((128, 128), (64, 128), (28, 64)):
    x = torch.randn(n, c_in, device=dev, generator=g)
    ea = torch.randn(4 * n, 20, device=dev, generator=g)
    We, be = torch.randn(c_in, 20, device=dev) * .1, torch.randn(c_in, device=dev)
    Wj, Wi, bj = torch.randn(c_out, c_in, device=dev) * .1, torch.randn(c_out, c_in, device=dev) * .1, torch.randn(c_out, device=dev)
    sc, sh = torch.ones(c_out, device=dev), torch.zeros(c_out, device=dev)
    f = lambda: ops.sage_layer_fused_fwd(plan.rowptr, plan.src, n, x, ea, We, be, Wj, bj, Wi, sc, sh, True, eid=plan.eid)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(sys.argv[1]), c_in, c_out, "%.4f ms" % (e0.elapsed_time(e1) / 10))
