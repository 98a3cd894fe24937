import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from helpers import kf96_state_dict
from dgnn_amd import ops
from dgnn_amd._lib import lib, ptr
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
from dgnn_amd.graph import GraphPlan
DEV='cuda:0'
net=SurfaceNet(reconbench_pretrained(device=DEV)); net.load_state_dict(kf96_state_dict()); net=net.to(DEV).eval()
adj,_,_=delaunay_tet_graph(20000,3); n=adj.shape[0]//4
x=hashed_normal(np.arange(n),29,seed=1,device=DEV); ea=hashed_normal(np.arange(4*n),20,seed=2,device=DEV)
ei=torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
plan=GraphPlan(ei,n,n)
h=x[:,1:]
for i in range(2): h=net._eval_layers_one(i,h,ea,plan)
conv=net.convs[2][0]
eas=plan.sorted_edge_attr(ea)
a=ops.aggregate_fwd(plan.rowptr,plan.src,None,n,h,eas,conv.lin_e.weight,conv.lin_e.bias)
ref=net._eval_layers_one(2,h,ea,plan)
ops.GEMM_MODE=1
ntiles=(n+31)//32; AF=12416
for rep in range(4):
    hin=h.clone()
    dump=torch.zeros(ntiles*AF,dtype=torch.float32,device=DEV)
    lib().dgnn_debug_trace_buffer(ptr(dump), dump.numel()//2)
    o=net._eval_layers(hin,n,ea,[plan]*4,True,only=2)
    torch.cuda.synchronize(); lib().dgnn_debug_trace_buffer(None,0)
    bad=((o-ref).abs()>2e-3*ref.abs().max()).any(1).nonzero().flatten().tolist()
    raw=dump.view(torch.int16).view(ntiles,32,776)[:,:,:768].reshape(ntiles,32,32,3,8)   # [tile,row,octet,part,8]
    parts=(raw.to(torch.int32)<<16).view(torch.float32)
    A=parts.sum(3).reshape(ntiles*32,256)[:n]
    exp=torch.cat([a,hin],1)
    err=(A-exp).abs()
    print('rep',rep,'bad out rows',bad[:6])
    for i in bad[:3]:
        ea_=err[i,:128]; ex=err[i,128:]
        print('   row',i,'a-part max err',ea_.max().item(),'n wrong',int((ea_>1e-4*exp[i,:128].abs().max()).sum()),' x-part max err',ex.max().item(),'n wrong',int((ex>0).sum()),
              ' wrong a cols',(ea_>1e-4*exp[i,:128].abs().max()).nonzero().flatten()[:10].tolist())
        # which edge explains? recompute a with one edge's phi*x replaced etc. skipped
