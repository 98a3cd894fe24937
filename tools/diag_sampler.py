"""Host time of one block build (4 hops, 2048 targets, 1M-tet scene), GPU idle before and after: python tools/diag_sampler.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgnn_amd.sampler import NeighborSampler
from dgnn_amd.synthetic import delaunay_tet_graph
dev = "cuda:0"
adj, _, _ = delaunay_tet_graph(150000, 0)
n = adj.shape[0] // 4
ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
idx = torch.randperm(n)[:2048 * 60].to(dev)
for tp in (False, True):
    s = NeighborSampler(ei, sizes=[-1] * 4, node_idx=idx, num_nodes=n, batch_size=2048, prefetch=False, transposed_plans=tp)
    s._side = torch.cuda.Stream(dev)
    ts = []
    for k in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if tp:
            out = s._build_on_side(idx[k * 2048:(k + 1) * 2048])
        else:
            out = s.sample(idx[k * 2048:(k + 1) * 2048])
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[5:])
    print("one_call=%s mailbox=%s transposed_plans=%s: host ms per block: median %.3f  min %.3f  max %.3f" % (
        os.environ.get("DGNN_KHOP_ONE_CALL", "1"), os.environ.get("DGNN_KHOP_MAILBOX", "1"), tp, ts[len(ts) // 2], ts[0], ts[-1]))
