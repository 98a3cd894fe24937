"""Who waits in the wave-specialised kernel: s_memtime ticks (100 MHz) spent in the hand-off polls, per role.  python tools/ws_timing.py"""
import sys, os, ctypes, numpy as np, torch
os.environ["DGNN_WS_NT"] = str(int(os.environ.get("DGNN_WS_NT", "1")) | 512)
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from dgnn_amd import ops
from dgnn_amd.graph import GraphPlan
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
from test_gpu_parity import hip_static
dev = "cuda:0"
adj, _, _ = delaunay_tet_graph(150000, 3)
n = adj.shape[0] // 4
ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
ea = hashed_normal(np.arange(4 * n), 20, seed=2, device="cpu").to(dev)
x = hashed_normal(np.arange(n), 128, seed=1, device="cpu").to(dev)
net = hip_static()
plan = GraphPlan(ei, n, n, hint=ops.PLAN_HINT_REFERENCE)
L = ops.lib()
buf = (ctypes.c_ulonglong * 8)()
fn = L.dgnn_ws_debug_read if hasattr(L, "dgnn_ws_debug_read") else ctypes.CDLL(os.path.join("dgnn_amd", "libdgnn_hip.so")).dgnn_ws_debug_read
for decode in (False, True):
    for _ in range(3):
        net._eval_layers(x, n, ea, [plan] * 4, True, only=3, decode=decode)
    fn(buf, 1)
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        net._eval_layers(x, n, ea, [plan] * 4, True, only=3, decode=decode)
    e1.record(); torch.cuda.synchronize()
    fn(buf, 1)
    v = [b / reps for b in buf]
    nw = 256 * 8
    print("decode %s: %.4f ms per launch (n %d); per producer wave: wait %.1f us of %.1f us; per consumer wave: wait %.1f us, main %.1f us of %.1f us" % (
        decode, e0.elapsed_time(e1) / reps, n, v[0] / nw / 100, v[1] / nw / 100, v[2] / nw / 100, v[4] / nw / 100, v[3] / nw / 100))
