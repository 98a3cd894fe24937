"""Is this box one of those that run the MFMA-heavy fp32 kernels at half speed?  Runs the fused 128->128 layer (fp32 I/O, default arithmetic) and
the bf16-storage one back to back for a few seconds each while a thread samples clock and power through rocm-smi; prints per-launch times over
time and the samples.  (about one box in four does; the bf16 kernels on the same box are unaffected.)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgnn_amd import ops
from dgnn_amd.graph import GraphPlan
from dgnn_amd.synthetic import delaunay_tet_graph
dev = "cuda:0"
adj, _, _ = delaunay_tet_graph(150000, 0)
n = adj.shape[0] // 4
plan = GraphPlan(torch.from_numpy(adj.T.astype(np.int64)).to(dev), n, n)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.relu(torch.randn(n, 128, device=dev, generator=g))
ea = torch.randn(4 * n, 20, device=dev, generator=g)
We, be = torch.randn(128, 20, device=dev) * .1, torch.randn(128, device=dev)
Wj, Wi, bj = torch.randn(128, 128, device=dev) * .1, torch.randn(128, 128, device=dev) * .1, torch.randn(128, device=dev)
sc, sh = torch.ones(128, device=dev), torch.zeros(128, device=dev)
xb = ops.cast_to_bf16(x)
f32 = lambda: ops.sage_layer_fused_fwd(plan.rowptr, plan.src, n, x, ea, We, be, Wj, bj, Wi, sc, sh, True, eid=plan.eid)
b16 = lambda: ops.sage_layer_fused_fwd_bf16(plan.rowptr, plan.src, n, xb, 128, ea, We, be, Wj, bj, Wi, sc, sh, True, eid=plan.eid)
samples, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [l.split(":")[-1].strip() for l in o.splitlines() if "sclk" in l or "Power (W)" in l]
            samples.append((time.perf_counter(), " ".join(s)))
        except Exception as e:  # noqa: BLE001
            samples.append((time.perf_counter(), repr(e)))
th = threading.Thread(target=sampler); th.start()
t00 = time.perf_counter()
for name, fn in (("fp32 128->128", f32), ("bf16 128->128", b16), ("fp32 again", f32)):
    fn(); torch.cuda.synchronize()
    times = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 4.0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        times.append((time.perf_counter() - t00, e0.elapsed_time(e1) / 50))
    print(name, " ".join("%.1fs:%.3f" % t for t in times[::max(1, len(times) // 10)]))
stop = True; th.join()
print("rocm-smi samples (s since start: sclk, power):", " | ".join("%.1f: %s" % (t - t00, s) for t, s in samples[::max(1, len(samples) // 16)]))
print(subprocess.run(["rocm-smi", "--showmaxpower", "--showperflevel"], capture_output=True, text=True).stdout.replace("\n", " ")[:600])
