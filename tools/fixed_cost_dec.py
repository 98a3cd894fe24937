"""Per-launch time of the last layer + decoder launch (dgnn_sage_layer_fused_decoder_fwd) against the plain 128 -> 128 layer at small n:
python tools/fixed_cost_dec.py"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgnn_amd import ops
from dgnn_amd.graph import GraphPlan
from dgnn_amd.synthetic import delaunay_tet_graph
dev = "cuda:0"
def t(f, reps=200):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for pts in (60, 2000, 10000, 20000):
    adj, _, _ = delaunay_tet_graph(pts, 0)
    n = adj.shape[0] // 4
    plan = GraphPlan(torch.from_numpy(adj.T.astype(np.int64)).to(dev), n, n)
    g = torch.Generator(device=dev).manual_seed(0)
    c = 128
    x = torch.relu(torch.randn(n, c, device=dev, generator=g)); ea = torch.randn(4 * n, 20, device=dev, generator=g)
    We, be = torch.randn(c, 20, device=dev) * .1, torch.randn(c, device=dev)
    Wj, Wi, bj = torch.randn(c, c, device=dev) * .1, torch.randn(c, c, device=dev) * .1, torch.randn(c, device=dev)
    sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    W0, b0, s1, h1 = torch.randn(64, c, device=dev) * .1, torch.randn(64, device=dev), torch.ones(64, device=dev), torch.zeros(64, device=dev)
    W3, b3 = torch.randn(2, 64, device=dev) * .1, torch.randn(2, device=dev)
    out = torch.empty(n, c, device=dev); lg = torch.empty(n, 2, device=dev)
    pp = ops.sage_layer_prepare(We, be, Wj, Wi)
    pd = ops.sage_layer_prepare(We, be, Wj, Wi, decoder=(W0, b0, s1, h1, W3, b3))
    plain = t(lambda: ops.sage_layer_fused_fwd(plan.rowptr, plan.src, n, x, ea, We, be, Wj, bj, Wi, sc, sh, True, out=out, eid=plan.eid, prepared=pp))
    dec = t(lambda: ops.sage_layer_fused_decoder_fwd(plan.rowptr, plan.src, n, x, ea, We, be, Wj, bj, Wi, sc, sh, True, W0, b0, s1, h1, W3, b3, out=lg, eid=plan.eid, prepared=pd))
    decu = t(lambda: ops.sage_layer_fused_decoder_fwd(plan.rowptr, plan.src, n, x, ea, We, be, Wj, bj, Wi, sc, sh, True, W0, b0, s1, h1, W3, b3, out=lg, eid=plan.eid))
    print("n=%7d  plain %.1f us   layer+decoder %.1f us (prepared), %.1f us (unprepared)" % (n, plain, dec, decu))
