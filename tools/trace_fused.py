"""Phase timeline of the fused layer kernel (workgroup 0), via dgnn_debug_trace_buffer."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgnn_amd._lib as _L
if len(sys.argv) > 2:
    _L.LIB_PATH = os.path.abspath(sys.argv[2])   # e.g. a what-if build
from dgnn_amd import ops
from dgnn_amd._lib import lib, ptr
from dgnn_amd.graph import GraphPlan
from dgnn_amd.synthetic import delaunay_tet_graph

dev = "cuda:0"
adj, _, _ = delaunay_tet_graph(int(sys.argv[1]) if len(sys.argv) > 1 else 150000, 0)
n = adj.shape[0] // 4
ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
plan = GraphPlan(ei, n, n)
c_in, c_out = (int(v) for v in os.environ.get("SHAPE", "128,128").split(","))      # SHAPE=28,64 / 64,128: the first two layers
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, c_in, device=dev, generator=g)
ea = torch.randn(4 * n, 20, device=dev, generator=g)
We, be = torch.randn(c_in, 20, device=dev) * .1, torch.randn(c_in, device=dev)
Wj, Wi, bj = torch.randn(c_out, c_in, device=dev) * .1, torch.randn(c_out, c_in, device=dev) * .1, torch.randn(c_out, device=dev)
sc, sh = torch.ones(c_out, device=dev), torch.zeros(c_out, device=dev)
f = lambda: ops.sage_layer_fused_fwd(plan.rowptr, plan.src, n, x, ea, We, be, Wj, bj, Wi, sc, sh, True)
f(); torch.cuda.synchronize()
NT = 40
buf = torch.zeros(NT * 12 * 8, dtype=torch.int64, device=dev)
lib().dgnn_debug_trace_buffer(ptr(buf), buf.numel())
f(); torch.cuda.synchronize()
lib().dgnn_debug_trace_buffer(None, 0)
t = buf.cpu().numpy().reshape(NT, 12, 8).astype(np.float64)
t0 = t[t > 0].min()
t = np.where(t > 0, (t - t0) / 100.0, np.nan)  # wall_clock64 ticks at 100 MHz -> microseconds
np.set_printoptions(linewidth=200, precision=2, suppress=True)
# uniform kernel phases: 0 P start, 1 loads landed + idx advanced, 2 rows written, 3 next loads issued, 4 barrier released, 5 MFMA done
sl = slice(5, 35)
print("tile period (us): %.2f" % np.nanmean(np.diff(t[sl, 0, 4])))
for w in (0, 3, 4, 7):
    ep = np.nanmean(t[sl, w, 6] - t[sl, w, 4]) if np.isfinite(t[sl, w, 6]).any() else float("nan")   # phase 6: delayed epilogue done (fused_mfma.hip)
    print("w%d: wait-loads %.2f  gather-compute %.2f  issue-next %.2f  barrier wait %.2f  epilogue+mfma %.2f (epilogue %.2f)  next P starts %.2f after" % (
        w, np.nanmean(t[sl, w, 1] - t[sl, w, 0]), np.nanmean(t[sl, w, 2] - t[sl, w, 1]), np.nanmean(t[sl, w, 3] - t[sl, w, 2]),
        np.nanmean(t[sl, w, 4] - t[sl, w, 3]), np.nanmean(t[sl, w, 5] - t[sl, w, 4]), ep, np.nanmean(t[sl, w, 0][1:] - t[sl, w, 5][:-1])))
