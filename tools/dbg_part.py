"""Flake hunt: partitioned-vs-whole bit identity, per layer, repeated."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from dgnn_amd.config import Config
from dgnn_amd.graph import GraphPlan
from dgnn_amd.partition import build_local_part, rcb_partition
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
from test_gpu_parity import hip_static
DEV = 'cuda:0'
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
adj, cent, _ = delaunay_tet_graph(4000, seed=6)
n = adj.shape[0] // 4
ei = adj.T.astype(np.int64)
x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
net = hip_static()
eit = torch.from_numpy(ei).to(DEV)
wplan = GraphPlan(eit, n, n)
part = rcb_partition(cent, world)
lps = [build_local_part(ei, part, r, world) for r in range(world)]
plans = [GraphPlan(torch.from_numpy(lp.edge_index).to(DEV), lp.n_own + lp.n_halo, lp.n_own) for lp in lps]
eas = [ea[torch.from_numpy(lp.edge_gid).to(DEV)] for lp in lps]
own = [torch.from_numpy(lp.own_gid).to(DEV) for lp in lps]
halo = [torch.from_numpy(lp.halo_gid).to(DEV) for lp in lps]
nbad = 0
for rep in range(reps):
    h = x[:, 1:]
    hs = [x[torch.cat([own[r], halo[r]])][:, 1:] for r in range(world)]
    for i in range(net.num_layers):
        f1 = net._eval_layers(h, n, ea, [wplan] * 4, True, only=i)
        f2 = net._eval_layers(h, n, ea, [wplan] * 4, True, only=i)
        if not torch.equal(f1, f2):
            rows = (f1 != f2).any(1).nonzero().flatten()
            print('rep', rep, 'layer', i, 'WHOLE run-to-run differs rows', rows[:8].tolist(), len(rows), (f1 - f2).abs().max().item())
            nbad += 1
        outs = [net._eval_layers(hs[r], lps[r].n_own, eas[r], [plans[r]] * 4, True, only=i) for r in range(world)]
        glob = torch.empty_like(f1)
        for r in range(world):
            glob[own[r]] = outs[r]
        if not torch.equal(glob, f1):
            rows = (glob != f1).any(1).nonzero().flatten()
            cols = (glob != f1).any(0).nonzero().flatten()
            print('rep', rep, 'layer', i, 'PART differs rows', rows[:8].tolist(), len(rows), 'cols', cols[:8].tolist(), len(cols),
                  'max', (glob - f1).abs().max().item(), 'owner', part[rows[:8].cpu().numpy()].tolist(),
                  'local', [int((own[part[int(g)]] == int(g)).nonzero()) for g in rows[:8].tolist()])
            nbad += 1
        h = f1
        hs = [torch.cat([f1[own[r]], f1[halo[r]]]) for r in range(world)]
print('done', reps, 'bad events', nbad)
