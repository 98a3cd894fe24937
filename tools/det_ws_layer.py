"""Where a run-to-run difference of the wave-specialised 128 -> 128 layer sits: one layer on the Ignatius graph, N launches, the cells / channels / ratios that
differ from the first launch.  python tools/det_ws_layer.py [repeats]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from dgnn_amd.graph import GraphPlan
from dgnn_amd.synthetic import hashed_normal
from test_gpu_parity import hip_static
from helpers import gold
dev = "cuda:0"
g = gold("static_f4_ignatius_full.npz")
n = g["x"].shape[0]
pairs = np.stack([np.repeat(np.arange(n, dtype=np.int64), 4), g["adj_dst"].astype(np.int64)], 1)
ei = torch.from_numpy(pairs).to(dev).t().contiguous()
ea = hashed_normal(np.arange(4 * n), 20, seed=2, device="cpu").to(dev)
# rows of very different magnitude from cell to cell: the per-row power-of-two scale changes between tiles
x = (hashed_normal(np.arange(n), 128, seed=1, device="cpu") * torch.exp2(torch.randint(-6, 7, (n, 1), generator=torch.Generator().manual_seed(3)).float())).to(dev)
net = hip_static()
plan = GraphPlan(ei, n, n)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = net._eval_layers(x.clone(), n, ea, [plan] * 4, True, only=1).clone()
events = 0
for i in range(reps):
    o = net._eval_layers(x.clone(), n, ea, [plan] * 4, True, only=1)
    d = (o != first)
    k = d.any(1).nonzero().flatten()
    if k.numel():
        events += 1
        if events <= 8:
            for c in k[:6].tolist():
                ch = d[c].nonzero().flatten().tolist()
                r = (o[c][ch] / first[c][ch]).tolist()
                print("rep %d cell %d (tile %d row %d) channels %s..%s (%d) ratio %s" % (i, c, c // 32, c % 32, ch[0], ch[-1], len(ch), [round(v, 4) for v in r[:4]]))
print("WS=%s RING=%s: %d of %d launches differ" % (os.environ.get("DGNN_WS", "1"), os.environ.get("DGNN_WS_RING", "22"), events, reps))
