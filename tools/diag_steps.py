import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from dgnn_amd import ops
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.graph import GraphPlan
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
dev="cuda:0"
adj,_,x,ea = bench.make_scene(int(sys.argv[1]),0)
n=adj.shape[0]//4
net=SurfaceNet(reconbench_pretrained(device=dev)); net.load_state_dict(bench.load_weights()); net=net.to(dev).eval()
data=Config(x=x.to(dev),edge_attr=ea.to(dev),edge_index=torch.from_numpy(adj.T.astype(np.int64)).to(dev))
def step():
    plan=GraphPlan(data.edge_index,n,n,hint=ops.PLAN_HINT_REFERENCE)
    return net.inference_layer(data,plan=plan)
for i in range(8):
    torch.cuda.synchronize(); t0=time.perf_counter(); step(); torch.cuda.synchronize(); print("step",i,"%.2f ms"%((time.perf_counter()-t0)*1e3), "alloc %.1f GB reserved %.1f GB"%(torch.cuda.memory_allocated()/1e9, torch.cuda.memory_reserved()/1e9))
torch.cuda.synchronize(); t0=time.perf_counter()
for i in range(5): step()
torch.cuda.synchronize(); print("5 steps back to back: %.2f ms/step"%((time.perf_counter()-t0)/5*1e3))
