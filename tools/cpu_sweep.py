import sys, time, os
sys.path.insert(0, '.')
import torch, numpy as np
from bench import make_scene, load_weights
from dgnn_amd.config import Config, reconbench_pretrained
from oracle.static_edge_filters import SurfaceNet
adj, _, x, ea = make_scene(30000, 0)
n = adj.shape[0] // 4
net = SurfaceNet(reconbench_pretrained(device="cpu")); net.load_state_dict(load_weights()); net.eval()
data = Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(adj.T.astype(np.int64)))
print("cores", os.cpu_count())
for t in (1, 8, 16, 32, 64, 128):
    torch.set_num_threads(t)
    with torch.no_grad():
        net.inference_layer(data)
        t0 = time.perf_counter(); net.inference_layer(data); dt = time.perf_counter() - t0
    print(t, "threads:", round(n / dt), "tets/s")
