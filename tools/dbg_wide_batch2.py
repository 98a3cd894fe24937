"""debug: inference_batch_layer (k-hop blocks) at wide widths -- which layer / batch produces the wrong rows"""
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
h = x[:, 1:]
whole = []
for i in range(4):
    y = net._eval_layers(h, n, ea, [plan] * 4, True, only=i)
    whole.append(y.float() if isinstance(y, ops.SplitRows) else y)
    h = y
full = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=ei))
for bsz in (256, 64, 100, 31):
    sel = torch.arange(0, n, 7, device=DEV)[:bsz]
    loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=sel, num_nodes=n, batch_size=bsz, shuffle=False)
    for batch_size, n_id, adjs in loader:
        n_id = n_id.to(DEV)
        xb = ops.gather_rows(x[:, 1:].contiguous(), n_id.to(torch.int32))
        for i in range(4):
            e_idx, e_id, size = adjs[i]
            p = plan_for(e_idx.to(DEV), size[0], size[1], hint=ops.PLAN_HINT_GROUPED)
            eab = ops.gather_rows(ea, e_id.to(DEV).to(torch.int32))
            xb = net._eval_layers(xb, p.n_dst, eab, [p] * 4, True, only=i)
            yf = xb.float() if isinstance(xb, ops.SplitRows) else xb
            ref = whole[i][n_id[:p.n_dst]]
            d = (yf - ref).abs()
            bad = (d.max(dim=1).values > 1e-3 * max(1.0, float(whole[i].abs().max()))).nonzero().flatten()
            print("bsz", bsz, "layer", i, "size", tuple(size), type(xb).__name__, "max diff %.3e" % float(d.max()), "bad rows", bad.numel(),
                  (int(bad.min()), int(bad.max())) if bad.numel() else "", "finite", bool(torch.isfinite(yf).all()))
        lg = net._eval_decoder(xb)
        d = (lg - full[n_id[:batch_size]]).abs().max(dim=1).values
        bad = (d > 1e-3).nonzero().flatten()
        print("bsz", bsz, "decoder M", lg.size(0), "max diff %.3e" % float(d.max()), "bad rows", bad.numel(), (int(bad.min()), int(bad.max())) if bad.numel() else "")
        # the decoder on the SAME rows padded to 256 (what a whole tile sees)
        if isinstance(xb, ops.SplitRows):
            xf = xb.float()
            lg2 = net._eval_decoder(xf)
            print("   decoder on the fp32 rows: max diff vs full %.3e" % float((lg2 - full[n_id[:batch_size]]).abs().max()))
