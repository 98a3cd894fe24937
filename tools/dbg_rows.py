import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from helpers import kf96_state_dict
from dgnn_amd import ops
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
from dgnn_amd.graph import GraphPlan
DEV='cuda:0'
net=SurfaceNet(reconbench_pretrained(device=DEV)); net.load_state_dict(kf96_state_dict()); net=net.to(DEV).eval()
P=int(sys.argv[1]) if len(sys.argv)>1 else 1500
adj,_,_=delaunay_tet_graph(P,3); n=adj.shape[0]//4
x=hashed_normal(np.arange(n),29,seed=1,device=DEV); ea=hashed_normal(np.arange(4*n),20,seed=2,device=DEV)
ei=torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
plan=GraphPlan(ei,n,n)
h=x[:,1:]
acts=[h]
for i in range(4):
    h=net._eval_layers_one(i,h,ea,plan); acts.append(h)
for mode in (0,1,2,3,4):
    ops.GEMM_MODE=mode
    for i in range(4):
        tot=0; pos=[]
        for rep in range(20):
            hin=acts[i].clone() if i>0 else acts[i]
            o=net._eval_layers(hin,n,ea,[plan]*4,True,only=i)
            ref=acts[i+1]
            bad=((o-ref).abs()>2e-3*ref.abs().max()).any(1).nonzero().flatten()
            tot+=len(bad); pos+= [(int(b)%64, int(b)//32) for b in bad[:4]]
        print('mode',mode,'layer',i,'bad rows in 20 runs:',tot, pos[:10])
