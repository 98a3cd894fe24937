"""BASELINE config 3 as a FUNCTIONAL check (missing 3): the Updated model at ModelNet10's widths and batch size
(reference configs/modelnet.yaml:44,56: [128,256,512,1024], batch 1024; model learning/surfaceNetUpdatedEdgeFilters.py:216-251,
step learning/runModel.py:264-282) trained for K Adam steps twice -- fp32 storage and bf16 storage -- from the SAME initial
weights on the SAME sequence of 4-hop blocks of one synthetic scene with learnable soft-occupancy targets.  The bf16-storage
gradients are 5-9 % rms away from fp64 at the first layers (ReLU-mask flips); whether the format is
usable is a question about the LOSS CURVE, which this answers: the two smoothed curves are compared step by step.

    python tools/config3_convergence.py [--steps 300] [--widths 128,256,512,1024] [--batch 1024] [--points 10000]

prints one JSON line: both curves (mean of every `--window` steps), the largest relative gap between them, the final losses."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def run(widths=(128, 256, 512, 1024), batch=1024, steps=300, points=10000, window=25, seed=0, dev="cuda:0", lr=0.005):
    from dgnn_amd import functional as Fn
    from dgnn_amd.config import Config
    from dgnn_amd.learning.runModel import Trainer, make_adam
    from dgnn_amd.learning.surfaceNetUpdatedEdgeFilters import SurfaceNet as UpdatedNet
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal

    adj, _, _ = delaunay_tet_graph(points, seed)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
    x = hashed_normal(np.arange(n), 29, seed=1, device=dev)
    x[:, 0] = x[:, 0].abs() + 0.05                                  # column 0: the cell volume (loss weight, runModel.py:193-199)
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=dev)
    # targets the network can learn from a cell's own features AND its neighbours' (so the message passing matters): soft occupancy
    nb = x[torch.from_numpy(adj[:, 1].reshape(n, 4).astype(np.int64)).to(dev).clamp_(0, n - 1), 5].mean(1, keepdim=True)
    occ = torch.sigmoid(2.0 * x[:, 3:4] + x[:, 7:8] + 1.5 * nb)
    y = torch.cat([occ, 1 - occ], 1)
    g = torch.Generator().manual_seed(seed)
    per = (n // batch) * batch
    need = batch * steps
    idx = torch.cat([torch.randperm(n, generator=g)[:per] for _ in range(need // per + 1)])[:need].to(dev)
    uclf = Config.wrap(dict(training=dict(model_params=list(widths), model_name="sage+", loss="kl"),
                            features=dict(normalization_feature=1, keep_normalization_feature=0), temp=dict(device=dev)))
    torch.manual_seed(seed)
    sd0 = {k: v.clone() for k, v in UpdatedNet(28, uclf).state_dict().items()}
    curves = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        net = UpdatedNet(28, uclf)
        net.load_state_dict(sd0)
        net = net.to(dev).train()
        net.set_storage_dtype(dt)
        opt = make_adam(net.parameters(), lr)
        loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=idx, num_nodes=n, batch_size=batch, shuffle=False)
        losses = []
        for bs, n_id, adjs in loader:
            ids = n_id[:bs]
            by, vol = y[ids], x[ids, 0]

            def loss_fn(logits):          # the Trainer's fused loss (runModel.py:171-209) and its gradient
                got = Fn.ops.kl_cell_loss_step(logits, by, vol, 0)
                return got[0], got[2]
            loss = net.train_step_direct(Config(x=x, edge_attr=ea, n_id=n_id, adjs=adjs), loss_fn)
            if loss is None:              # (a configuration the one-call form does not take: the autograd path)
                opt.zero_grad()
                logits = net(Config(x=x, edge_attr=ea, n_id=n_id, adjs=adjs)).float()
                loss, _ = Fn.kl_cell_loss(logits, by, vol)
                loss.backward()
                loss = loss.detach()
            opt.step()
            losses.append(loss)
        curves[name] = torch.stack(losses).double().cpu().numpy()
        del net, opt, loader
    k = (steps // window) * window
    a = curves["f32"][:k].reshape(-1, window).mean(1)
    b = curves["bf16"][:k].reshape(-1, window).mean(1)
    gap = np.abs(b - a) / a
    return dict(widths=list(widths), batch=batch, steps=steps, window=window, scene_tets=n, lr=lr,
                loss_f32=[round(float(v), 6) for v in a], loss_bf16=[round(float(v), 6) for v in b],
                loss_f32_first=float(a[0]), loss_f32_last=float(a[-1]), loss_bf16_last=float(b[-1]),
                max_rel_gap=float(gap.max()), last_rel_gap=float(gap[-1]),
                # the first 100 steps are the steep part of the curve (the loss falls 4-5 x per 25 steps there: a lag of three steps is a 20 % gap)
                max_rel_gap_early=float(gap[:max(1, 100 // window)].max()), max_rel_gap_late=float(gap[max(1, 100 // window):].max()) if k > 100 else 0.0, finite=bool(np.isfinite(curves["bf16"]).all() and np.isfinite(curves["f32"]).all()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--widths", type=str, default="128,256,512,1024")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--points", type=int, default=10000)
    ap.add_argument("--window", type=int, default=25)
    args = ap.parse_args()
    print(json.dumps(run(tuple(int(v) for v in args.widths.split(",")), args.batch, args.steps, args.points, args.window)))
