"""Overhead check of the partitioned orchestration: one rank owning the whole 1M-tet scene (no halo) vs bench.py's N=1 path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from dgnn_amd.partition import PartitionedScene
from test_gpu_parity import hip_static
net = hip_static()
scene = PartitionedScene.build_synthetic(150000, 0, 0, 1, "cuda:0")
for _ in range(5):
    scene.inference_layer(net)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    scene.inference_layer(net)
torch.cuda.synchronize()
print("partitioned world=1: %.3f ms/step, n_own %d" % ((time.perf_counter() - t) / 20 * 1e3, scene.n_own))
