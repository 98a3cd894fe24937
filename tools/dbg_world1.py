import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from helpers import kf96_state_dict
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.partition import PartitionedScene
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
from dgnn_amd.graph import GraphPlan
DEV='cuda:0'
net=SurfaceNet(reconbench_pretrained(device=DEV)); net.load_state_dict(kf96_state_dict()); net=net.to(DEV).eval()
scene=PartitionedScene.build_synthetic(1500,3,0,1,DEV)
adj,_,_=delaunay_tet_graph(1500,3); n=adj.shape[0]//4
data=Config(x=hashed_normal(np.arange(n),29,seed=1,device=DEV), edge_attr=hashed_normal(np.arange(4*n),20,seed=2,device=DEV), edge_index=torch.from_numpy(adj.T.astype(np.int64)).to(DEV))
a=net.inference_layer(data); b=net.inference_layer(data); print('repeat equal', torch.equal(a,b))
c=scene.inference_layer(net); print('scene equal', torch.equal(a,c), (a-c).abs().max().item())
print('inputs equal', torch.equal(scene.x_local, data.x), torch.equal(scene.edge_attr, data.edge_attr), torch.equal(scene.edge_index, data.edge_index), scene.n_halo)
plan=GraphPlan(data.edge_index,n,n)
h1=data.x[:,1:]; h2=scene.x_local[:,1:]
for i in range(4):
    o1=net._eval_layers(h1,n,data.edge_attr,[plan]*4,True,only=i)
    o2=net._eval_layers(h2,n,scene.edge_attr,[scene.plan]*4,True,only=i)
    buf=torch.empty((n,o2.size(1)),device=DEV); buf[:n]=o2
    print('layer',i,'equal',torch.equal(o1,o2),(o1-o2).abs().max().item(), 'ptr align', h1.data_ptr()%256, h2.data_ptr()%256)
    h1=o1; h2=buf
print("---- layer-2 repeatability on real activations")
h=data.x[:,1:]
for i in range(2): h=net._eval_layers(h,n,data.edge_attr,[plan]*4,True,only=i)
from dgnn_amd import ops
for mode in (0,1):
    ops.GEMM_MODE=mode
    outs=[net._eval_layers(h,n,data.edge_attr,[plan]*4,True,only=2) for _ in range(6)]
    for k in range(1,6):
        bad=(outs[k]!=outs[0]).any(1).nonzero().flatten()
        print('mode',mode,'run',k,'rows differing',len(bad), bad[:12].tolist(), 'maxdiff', (outs[k]-outs[0]).abs().max().item())
print('nan in h?', torch.isnan(h).any().item(), 'h zeros frac', (h==0).float().mean().item())
print("---- which is right? compare against the unfused path")
ops.GEMM_MODE=1
h1=data.x[:,1:]
for i in range(4):
    ref=net._eval_layers_one(i,h1,data.edge_attr,plan)
    o1=net._eval_layers(h1,n,data.edge_attr,[plan]*4,True,only=i)
    hb=torch.empty_like(h1.contiguous()) if i>0 else h1
    if i>0: hb.copy_(h1)
    o2=net._eval_layers(hb,n,data.edge_attr,[plan]*4,True,only=i)
    e1=(o1-ref).abs().max().item(); e2=(o2-ref).abs().max().item()
    bad=((o1-ref).abs()>1e-3).any(1).nonzero().flatten()
    print('layer',i,'err direct',e1,'err copy',e2,'bad rows',len(bad),bad[:16].tolist(), 'h1 ptr', hex(h1.data_ptr()), 'hb ptr', hex(hb.data_ptr()))
    h1=ref
print("---- pattern of wrong entries")
bad=((o2-ref).abs()>1e-3)
rows=bad.any(1).nonzero().flatten(); cols=bad.any(0).nonzero().flatten()
print('n bad rows',len(rows),'first',rows[:20].tolist()); print('bad cols',cols.tolist()[:40])
print('rows mod 32', sorted(set((rows%32).tolist()))[:40])
tile=(rows//32); print('tiles', sorted(set(tile.tolist()))[:20])
r0=rows[0].item(); print('row',r0,'o2',o2[r0,:8].tolist(),'ref',ref[r0,:8].tolist())
