import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from dgnn_amd import ops
from dgnn_amd.graph import GraphPlan
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
dev='cuda:0'
adj,_,_=delaunay_tet_graph(1500,3); n=adj.shape[0]//4
ei=torch.from_numpy(adj.T.astype(np.int64)).to(dev)
plan=GraphPlan(ei,n,n)
g=torch.Generator(device=dev).manual_seed(0)
for (ci,co) in ((28,64),(64,128),(128,128)):
    x=torch.randn(n,ci,device=dev,generator=g); ea=torch.randn(4*n,20,device=dev,generator=g)
    We,be=torch.randn(ci,20,device=dev)*.1,torch.randn(ci,device=dev)
    Wj,Wi,bj=torch.randn(co,ci,device=dev)*.1,torch.randn(co,ci,device=dev)*.1,torch.randn(co,device=dev)
    sc,sh=torch.ones(co,device=dev),torch.zeros(co,device=dev)
    eas=plan.sorted_edge_attr(ea)
    for mode in (0,1):
        outs=[ops.sage_layer_fused_fwd(plan.rowptr,plan.src,n,x,eas,We,be,Wj,bj,Wi,sc,sh,True,gemm_mode=mode) for _ in range(5)]
        d=[(o-outs[0]).abs().max().item() for o in outs]
        bad=(outs[1]!=outs[0]).any(1).nonzero().flatten()
        print(ci,co,'mode',mode,'run-to-run max diff',d, 'rows differing', bad[:10].tolist(), len(bad))
