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
a_ref=ops.aggregate_fwd(plan.rowptr,plan.src,None,n,h,eas,conv.lin_e.weight,conv.lin_e.bias)
ref=net._eval_layers_one(2,h,ea,plan)
ops.GEMM_MODE=1
for rep in range(3):
    hin=h.clone()
    dbg=torch.zeros(n,256,device=DEV)
    lib().dgnn_debug_trace_buffer(ptr(dbg), dbg.numel()//2)
    o=net._eval_layers(hin,n,ea,[plan]*4,True,only=2)
    torch.cuda.synchronize(); lib().dgnn_debug_trace_buffer(None,0)
    bad=((o-ref).abs()>2e-3*ref.abs().max()).any(1).nonzero().flatten()
    ea_=(dbg[:,:128]-a_ref).abs().max(1).values; ex=(dbg[:,128:]-hin).abs().max(1).values
    print('rep',rep,'bad out rows',bad[:8].tolist(),' rows with bad a:',(ea_>1e-3).nonzero().flatten()[:8].tolist(),' bad xd:',(ex>0).nonzero().flatten()[:8].tolist())
    for b in bad[:2].tolist():
        print('   row',b,'a err',ea_[b].item(),'xd err',ex[b].item(), 'out err',(o[b]-ref[b]).abs().max().item())
