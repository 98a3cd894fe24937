"""Cell orders compared on the metric graph: the loader's Morton order against a Hilbert order of the same centroids (numpy, Skilling's transform),
and Morton keys of other widths -- step time of inference_layer on the relabelled scene.   python tools/order_experiment.py [points]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from dgnn_amd import ops
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.graph import GraphPlan
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.processing.reorder import reorder_edges, cell_order_morton

dev = "cuda:0"
points = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
adj, cent, x, ea = bench.make_scene(points, 0)
n = adj.shape[0] // 4
net = SurfaceNet(reconbench_pretrained(device=dev, convs=(64, 128, 128, 128)))
net.load_state_dict(bench.load_weights())
net = net.to(dev).eval()
data0 = Config(x=x.to(dev), edge_attr=ea.to(dev), edge_index=torch.from_numpy(adj.T.astype(np.int64)).to(dev))


def hilbert_keys(c, bits=16):
    lo, hi = c.min(0), c.max(0)
    q = np.minimum(((c - lo) / np.maximum(hi - lo, 1e-30) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    X = [q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy()]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = np.where(m, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t
    key = np.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << 1) | ((X[i] >> b) & 1)
    return key


def run(order_np, label):
    order = torch.from_numpy(order_np.astype(np.int32)).to(dev)
    rank = torch.empty_like(order)
    rank[order.long()] = torch.arange(n, dtype=torch.int32, device=dev)
    ei, rows = reorder_edges(data0.edge_index, order, rank)
    d = Config(x=ops.gather_rows(data0.x, order), edge_attr=ops.gather_rows(data0.edge_attr, rows), edge_index=ei)
    step = lambda: net.inference_layer(d, plan=GraphPlan(d.edge_index, n, n, hint=ops.PLAN_HINT_REFERENCE))
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    dd = (ei[0] - ei[1]).abs().float()
    print("%-28s %.4f ms/step (%.4f .. %.4f)  %.3e tets/s   |src-dst| median %d  90%% %d  99%% %d" % (
        label, np.median(ts), min(ts), max(ts), n / np.median(ts) * 1e3, dd.median().item(), dd.quantile(0.9).item() if dd.numel() < 16e6 else -1,
        dd.kthvalue(int(0.99 * dd.numel())).values.item()), flush=True)


run(np.arange(n), "generator order")
o_m, _ = cell_order_morton(torch.from_numpy(cent).to(dev))
run(o_m.cpu().numpy(), "morton (library, 16 bits/axis)")
run(np.argsort(hilbert_keys(cent, 16), kind="stable"), "hilbert 16 bits/axis")
run(np.argsort(hilbert_keys(cent, 10), kind="stable"), "hilbert 10 bits/axis")
run(o_m.cpu().numpy(), "morton again")
