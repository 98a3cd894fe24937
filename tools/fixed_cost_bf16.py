"""Fixed cost of a bf16-storage fused-layer launch (k_sage_fused_bf16): time per launch at n = 330 .. 134k tets: python tools/fixed_cost_bf16.py"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgnn_amd import ops
from dgnn_amd.graph import GraphPlan
from dgnn_amd.synthetic import delaunay_tet_graph
dev = "cuda:0"
for pts in (60, 2000, 20000):
    adj, _, _ = delaunay_tet_graph(pts, 0)
    n = adj.shape[0] // 4
    plan = GraphPlan(torch.from_numpy(adj.T.astype(np.int64)).to(dev), n, n)
    for c_in, c_out in ((64, 128), (128, 128)):
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.relu(torch.randn(n, c_in, device=dev, generator=g)).to(torch.bfloat16); ea = torch.randn(4 * n, 20, device=dev, generator=g)
        We, be = torch.randn(c_in, 20, device=dev) * .1, torch.randn(c_in, device=dev)
        Wj, Wi, bj = torch.randn(c_out, c_in, device=dev) * .1, torch.randn(c_out, c_in, device=dev) * .1, torch.randn(c_out, device=dev)
        sc, sh = torch.ones(c_out, device=dev), torch.zeros(c_out, device=dev)
        f = lambda: ops.sage_layer_fused_fwd_bf16(plan.rowptr, plan.src, n, x, c_in, ea, We, be, Wj, bj, Wi, sc, sh, True, eid=plan.eid)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        print("n=%7d  %3d->%3d  %.1f us per launch" % (n, c_in, c_out, e0.elapsed_time(e1) / reps * 1e3))
