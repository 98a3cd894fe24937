"""Where a training step's time goes: host enqueue time and GPU time per phase (block builder / forward / loss / backward /
Adam), measured two ways -- phases separated by synchronize (GPU time per phase, events) and back to back (host time per
phase while the GPU runs behind).  python tools/diag_train.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.learning.runModel import Metrics, Trainer, adjust_learning_rate
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.sampler import NeighborSampler
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
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
opt = torch.optim.Adam(net.parameters(), lr=clf.training.learning_rate, fused=True)
adjust_learning_rate(opt, clf)
idx = torch.randperm(n)[:2048 * (2 * steps + 10)].to(dev)
PREFETCH = {"0": False, "1": True, "thread": "thread"}[os.environ.get("PREFETCH", "1")]
loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=idx, num_nodes=n, batch_size=2048, prefetch=PREFETCH)
it = iter(loader)
names = ["sample", "forward+loss", "backward", "adam"]


def one(sync_between, acc_host, acc_gpu):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    t = [time.perf_counter()]
    ev[0].record()
    bs, n_id, adjs = next(it)
    if sync_between: torch.cuda.synchronize()
    t.append(time.perf_counter()); ev[1].record()
    data = Config(all=all_, batch_n_id=n_id, batch_adjs=adjs)
    logits = net(data)
    ids = n_id[:adjs[-1].size[1]]
    data.batch_x = all_.x[ids]; data.batch_gt = all_.y[ids]
    loss = tr.calcLossAndOA(logits, None, data, clf, clf.training.metrics)
    opt.zero_grad()
    if sync_between: torch.cuda.synchronize()
    t.append(time.perf_counter()); ev[2].record()
    loss.backward()
    if sync_between: torch.cuda.synchronize()
    t.append(time.perf_counter()); ev[3].record()
    opt.step()
    if sync_between: torch.cuda.synchronize()
    t.append(time.perf_counter()); ev[4].record()
    if sync_between: torch.cuda.synchronize()
    for i in range(4):
        acc_host[i] += (t[i + 1] - t[i]) * 1e3
    return ev


for _ in range(5):
    one(False, [0] * 4, [0] * 4)
torch.cuda.synchronize()
for mode in (True, False):
    h, g = [0.0] * 4, [0.0] * 4
    evs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        evs.append(one(mode, h, g))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    for ev in evs:
        for i in range(4):
            g[i] += ev[i].elapsed_time(ev[i + 1])
    print("sync between phases" if mode else "back to back (no sync inside the step)", "wall %.3f ms/step" % wall)
    for i, nm in enumerate(names):
        print("   %-14s host %.3f ms   events %.3f ms" % (nm, h[i] / steps, g[i] / steps))
