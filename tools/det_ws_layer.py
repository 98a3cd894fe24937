"""Where a run-to-run difference of the wave-specialised last layer + decoder sits: synthetic Delaunay graph, N launches, the cells that differ from the majority
and their place in the workgroup's tile sequence.  python tools/det_ws_layer.py [points] [repeats] [decode]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from dgnn_amd.graph import GraphPlan
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
from test_gpu_parity import hip_static
dev = "cuda:0"
points = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
decode = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
adj, _, _ = delaunay_tet_graph(points, seed=points)
n = adj.shape[0] // 4
ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
ea = hashed_normal(np.arange(4 * n), 20, seed=2, device="cpu").to(dev)
x = hashed_normal(np.arange(n), 128, seed=1, device="cpu").to(dev)
net = hip_static()
plan = GraphPlan(ei, n, n)
ntiles = (n + 31) // 32
per = (ntiles + 7) // 8
nwg = min(ntiles, 256)
outs = [net._eval_layers(x, n, ea, [plan] * 4, True, only=3, decode=decode).clone() for _ in range(reps)]
ref = torch.stack(outs[:7]).median(0).values
ev = 0
for r, o in enumerate(outs):
    k = (o != ref).any(1).nonzero().flatten().tolist()
    if k:
        ev += 1
        if ev <= 12:
            tiles = sorted(set(c // 32 for c in k))
            for t in tiles[:3]:
                rows = [c % 32 for c in k if c // 32 == t]
                xcd = t // per; rel = t - xcd * per; wgx = (nwg + 7 - xcd) >> 3; it = rel // wgx; slot = rel % wgx
                t_hi = min(ntiles, (xcd + 1) * per); my_n = (t_hi - xcd * per - slot + wgx - 1) // wgx
                c = t * 32 + rows[0]
                print("  rep %d tile %d rows %s: xcd %d wg %d it %d of %d; now %s ref %s" % (r, t, rows, xcd, slot, it, my_n, o[c].tolist()[:2], ref[c].tolist()[:2]))
print("n %d tiles %d: %d of %d launches differ (decode %s, knobs %s)" % (n, ntiles, ev, reps, decode, os.environ.get("DGNN_WS_NT", "1")))
