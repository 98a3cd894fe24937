"""debug: inference_layer_batch at wide widths -- which layer / batch / path produces the wrong rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from dgnn_amd import ops
from dgnn_amd.config import Config
from dgnn_amd.graph import plan_for, GraphPlan
from dgnn_amd.sampler import NeighborSampler
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
from helpers import oracle_static
from test_gpu_parity import DEV, hip_static

convs = (64, 128, 256, 512)
adj, _, _ = delaunay_tet_graph(900, 6)
n = adj.shape[0] // 4
x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
onet = oracle_static(convs=convs, load=False, seed=4)
for m in onet.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.running_mean.normal_(0, 0.1)
        m.running_var.uniform_(0.5, 1.5)
net = hip_static(convs=convs, sd=onet.state_dict())
plan = GraphPlan(ei, n, n)
# whole-graph per-layer activations (fp32 rows), wide path on
h = x[:, 1:]
whole = []
for i in range(4):
    y = net._eval_layers(h, n, ea, [plan] * 4, True, only=i)
    whole.append(y.float() if isinstance(y, ops.SplitRows) else y)
    h = y
print("n =", n, "whole-graph layer maxima:", [float(w.abs().max()) for w in whole])
loader = NeighborSampler(ei, sizes=[-1], num_nodes=n, batch_size=1024, shuffle=False)
for mode in ("float", "float_nowide"):
    ops.WIDE_SR = mode != "float_nowide"
    x_all = x[:, 1:]
    for i in range(4):
        xs = []
        off = 0
        for bs, n_id, adj_ in loader:
            e_idx, e_id, size = adj_
            xb = ops.gather_rows(x_all, n_id.to(DEV).to(torch.int32))
            p = plan_for(e_idx.to(DEV), size[0], size[1], hint=ops.PLAN_HINT_GROUPED)
            eab = ops.gather_rows(ea, e_id.to(DEV).to(torch.int32))
            y = net._eval_layers(xb, p.n_dst, eab, [p] * 4, True, only=i)
            kind = type(y).__name__
            yf = y.float() if isinstance(y, ops.SplitRows) else y
            d = (yf - whole[i][off:off + bs]).abs()
            bad = (d.max(dim=1).values > 1e-3 * max(1.0, float(whole[i].abs().max()))).nonzero().flatten()
            print(mode, "layer", i, "batch@%d" % off, "size", tuple(size), kind, "max diff %.3e" % float(d.max()), "bad rows", bad.numel(),
                  (int(bad.min()), int(bad.max())) if bad.numel() else "")
            xs.append(yf)
            off += bs
        x_all = torch.cat(xs, 0)
        x_all = whole[i]       # continue from the correct rows: each layer judged on its own
ops.WIDE_SR = True
