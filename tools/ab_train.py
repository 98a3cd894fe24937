"""Interleaved A/B of training-step variants inside ONE process (box-to-box and run-to-run drift is +-15 %): segments of 40 steps
alternate between the settings, the median segment time per setting is reported.
    python tools/ab_train.py whole=1 whole=0 composite=0     all layers in one call | one call per layer | separate Functions
    python tools/ab_train.py aux=0 aux=1                      weight gradients on the second stream
    python tools/ab_train.py fused=3 fused=1 fused=0          fused launch chains (bit 0 backward, bit 1 forward statistics)
Every setting starts from the defaults (composite=1, whole=1, aux=0, fused=3); several flags: "whole=0,aux=1"."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgnn_amd import ops
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.learning.runModel import Metrics, Trainer, adjust_learning_rate
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.sampler import NeighborSampler
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
settings = sys.argv[1:] or ["whole=1", "whole=0"]
ROUNDS, SEG = 7, 40
dev = "cuda:0"
adj, _, _ = delaunay_tet_graph(150000, 0)
n = adj.shape[0] // 4
ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
x = hashed_normal(np.arange(n), 29, seed=1, device=dev); x[:, 0] = x[:, 0].abs() + 0.05
ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=dev)
occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
all_ = Config(x=x, y=torch.cat([occ, 1 - occ], 1), edge_attr=ea)
clf = reconbench_pretrained(device=dev); clf.temp.current_epoch = 0; clf.training.metrics = Metrics()
torch.manual_seed(0)
net = SurfaceNet(clf).to(dev).train()
tr = Trainer(net)
if os.environ.get("DGNN_TORCH_ADAM") == "1":
    opt = torch.optim.Adam(net.parameters(), lr=clf.training.learning_rate, fused=True)
else:
    from dgnn_amd.learning.runModel import make_adam
    opt = make_adam(net.parameters(), clf.training.learning_rate)
adjust_learning_rate(opt, clf)
per = (n // 2048) * 2048      # whole batches per permutation: no duplicate targets inside a batch
need = 2048 * (ROUNDS * len(settings) * SEG + 5 * len(settings) + 8)
idx = torch.cat([torch.randperm(n)[:per] for _ in range(need // per + 1)])[:need].to(dev)
it = iter(NeighborSampler(ei, sizes=[-1] * 4, node_idx=idx, num_nodes=n, batch_size=2048, reuse_buffers=os.environ.get("RING", "1") == "1"))
_wait = [0.0, 0]
_orig_finish = NeighborSampler._finish_regular
def _timed_finish(self, b):
    t0 = time.perf_counter(); out = _orig_finish(self, b); _wait[0] += time.perf_counter() - t0; _wait[1] += 1
    return out
NeighborSampler._finish_regular = _timed_finish
def apply(s):
    """every setting names the COMPLETE configuration (an earlier version of this tool changed one flag per setting and left the others
    where the previous setting had put them: 'whole=1 whole=0 composite=0' then measured composite=0 three times)"""
    from dgnn_amd._lib import lib
    ops.TRAIN_COMPOSITE, ops.TRAIN_WHOLE_MODEL = True, True
    lib().dgnn_train_set_aux_stream(0)
    lib().dgnn_train_set_fused(3)
    for part in s.split(","):
        k, v = part.split("=")
        if k == "whole": ops.TRAIN_WHOLE_MODEL = v == "1"
        elif k == "composite": ops.TRAIN_COMPOSITE = v == "1"
        elif k == "aux": lib().dgnn_train_set_aux_stream(int(v))
        elif k == "fused": lib().dgnn_train_set_fused(int(v))     # bit 0: backward chain, bit 1: statistics from the GEMM epilogue
        else: raise SystemExit("unknown setting " + s)
def seg(nsteps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(nsteps):
        bs, n_id, adjs = next(it)
        tr.train(Config(all=all_, batch_n_id=n_id, batch_adjs=adjs), opt, clf)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / nsteps * 1e3
for s in settings:
    apply(s); seg(5)
res = {s: [] for s in settings}
for r in range(ROUNDS):
    for s in settings:
        apply(s)
        res[s].append(seg(SEG))
print("time inside _finish_regular (join of the builder thread + views): %.3f ms per block" % (_wait[0] / max(_wait[1], 1) * 1e3))
for s in settings:
    v = sorted(res[s])
    print("%-14s median %.3f ms/step   min %.3f  max %.3f   (%s)" % (s, v[len(v) // 2], v[0], v[-1], " ".join("%.2f" % t for t in res[s])))
