"""cProfile of the Python side of SurfaceNet.inference_layer on a small scene (66k tets: the pass is host-bound there): python tools/prof_infer_py.py [points]"""
import cProfile, pstats, io, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.synthetic import delaunay_tet_graph
dev = "cuda:0"
adj, _, _ = delaunay_tet_graph(int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 0)
n = adj.shape[0] // 4
net = SurfaceNet(reconbench_pretrained(device=dev)); net.load_state_dict(bench.load_weights()); net = net.to(dev).eval()
g = torch.Generator().manual_seed(0)
ei = torch.from_numpy(np.ascontiguousarray(adj.astype(np.int64))).to(dev).t()
data = Config(x=torch.randn(n, 29, generator=g).to(dev), edge_attr=torch.randn(4 * n, 20, generator=g).to(dev), edge_index=ei)
for _ in range(20): net.inference_layer(data)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(500): net.inference_layer(data)
torch.cuda.synchronize()
print("n = %d tets: %.1f us per inference_layer call (wall)" % (n, (time.perf_counter() - t0) / 500 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): net.inference_layer(data)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
