"""Run-to-run determinism probe of the fused layers on the whole Ignatius scene (CGAL cell order: long, irregular gather latencies): N repeats of
inference_layer, rows that differ from the first run.  python tools/det_ws.py [repeats]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from dgnn_amd import ops
from dgnn_amd.config import Config
from test_gpu_parity import hip_static
from helpers import gold
dev = "cuda:0"
g = gold("static_f4_ignatius_full.npz")
n = g["x"].shape[0]
fg = np.random.default_rng(int(g["fgeom_seed"])).standard_normal((4 * n, 4)).astype(np.float32)
ea = torch.from_numpy(np.concatenate([fg, g["edge_attr16"]], axis=1)).to(dev)
pairs = np.stack([np.repeat(np.arange(n, dtype=np.int64), 4), g["adj_dst"].astype(np.int64)], 1)
data = Config(x=torch.from_numpy(g["x"]).to(dev), edge_attr=ea, edge_index=torch.from_numpy(pairs).to(dev).t())
net = hip_static()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = net.inference_layer(data).clone()
bad = []
for i in range(reps):
    o = net.inference_layer(data)
    k = (o != first).any(1).nonzero().flatten()
    if k.numel():
        bad.append((i, k.numel(), k[:3].tolist(), float((o - first).abs().max())))
print("WS=%s RING=%s FUSE_DEC=%s: %d of %d repeats differ" % (os.environ.get("DGNN_WS", "1"), os.environ.get("DGNN_WS_RING", "22"), os.environ.get("DGNN_FUSE_DECODER", "1"), len(bad), reps), bad[:6])
