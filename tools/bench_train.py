"""Training throughput ("fwd+bwd+Adam, reported separately"): the reference's training recipe
(configs/pretrained/reconbench.yaml: batch_size 2048 targets, 4-hop full-neighbour blocks, Adam, KL loss weighted by
cell volume) on the synthetic 150k-point scene, everything on the GPU: k-hop block builder (dgnn_amd.sampler) ->
SurfaceNet.forward (BN in train mode) -> loss -> backward through the HIP kernels -> Adam.
Prints one JSON line: supervised targets/s and block tets/s (all cells touched by the 4-hop blocks)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.learning.runModel import Metrics, Trainer, adjust_learning_rate
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.sampler import NeighborSampler
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal

dev = "cuda:0"
UPDATED = "--updated" in sys.argv   # surfaceNetUpdatedEdgeFilters ("sage+", edge embeddings chained layer to layer) instead of Static
sys.argv = [a for a in sys.argv if a != "--updated"]
points = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
adj, _, _ = delaunay_tet_graph(points, 0)
n = adj.shape[0] // 4
ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
x = hashed_normal(np.arange(n), 29, seed=1, device=dev)
x[:, 0] = x[:, 0].abs() + 0.05
ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=dev)
occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
all_ = Config(x=x, y=torch.cat([occ, 1 - occ], 1), edge_attr=ea)
clf = reconbench_pretrained(device=dev)
clf.temp.current_epoch = 0
clf.training.metrics = Metrics()
if UPDATED:
    import torch.nn.functional as F
    from dgnn_amd.learning.surfaceNetUpdatedEdgeFilters import SurfaceNet as UpdatedNet
    uclf = Config.wrap(dict(training=dict(model_params=[64, 128, 128, 128], model_name="sage+", loss="kl"),
                            features=dict(normalization_feature=1, keep_normalization_feature=0), temp=dict(device=dev)))
    net = UpdatedNet(28, uclf).to(dev).train()

    class _Tr:   # the reference's loss (runModel.py:171-211) on the Updated model's (x, edge_attr, n_id, adjs) batch layout
        def train(self, data, opt, clf):
            opt.zero_grad()
            ids = data.batch_n_id[:data.batch_adjs[-1][2][1]]
            logits = net(Config(x=data.all.x, edge_attr=data.all.edge_attr, n_id=data.batch_n_id, adjs=data.batch_adjs))
            w = data.all.x[ids, 0]
            loss = (F.kl_div(F.log_softmax(logits, dim=-1), data.all.y[ids], reduction="none").sum(1) * w).sum() / w.sum()
            loss.backward()
            opt.step()
            return loss
    tr = _Tr()
else:
    net = SurfaceNet(clf).to(dev).train()
    tr = Trainer(net)
opt = torch.optim.Adam(net.parameters(), lr=clf.training.learning_rate, fused=True)  # one launch for all 49 tensors
adjust_learning_rate(opt, clf)
g = torch.Generator().manual_seed(0)
idx = torch.randperm(n, generator=g)[:batch * (steps + 5)]
loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=idx.to(dev), num_nodes=n, batch_size=batch)
it = iter(loader)
block = 0
for _ in range(5):
    bs, n_id, adjs = next(it)
    tr.train(Config(all=all_, batch_n_id=n_id, batch_adjs=adjs), opt, clf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    bs, n_id, adjs = next(it)
    block += int(n_id.numel())
    loss = tr.train(Config(all=all_, batch_n_id=n_id, batch_adjs=adjs), opt, clf)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"metric": "training step (block builder + fwd + bwd + Adam), one MI355X", "model": "UpdatedEdgeFilters sage+" if UPDATED else "StaticEdgeFilters", "targets_per_s": round(batch * steps / dt, 1),
                  "block_tets_per_s": round(block / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "batch_targets": batch,
                  "avg_block_tets": round(block / steps, 1), "steps": steps, "scene_tets": n, "final_loss": float(loss)}))
