"""Training throughput ("fwd+bwd+Adam, reported separately"): the reference's training recipe
(configs/pretrained/reconbench.yaml: batch_size 2048 targets, 4-hop full-neighbour blocks, Adam, KL loss weighted by
cell volume) on the synthetic 150k-point scene, everything on the GPU: k-hop block builder (dgnn_amd.sampler) ->
SurfaceNet.forward (BN in train mode) -> loss -> backward through the HIP kernels -> Adam.

    python tools/bench_train.py [--updated] [--dtype bf16] [--points P] [--batch B] [--steps K]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_train.py --gpus N ...

--gpus N (BASELINE config 5): data-parallel replicas, ONE SCENE SHARD PER GPU (rank r trains on its own seeded scene),
weights broadcast from rank 0, one flat RCCL all-reduce of the gradients per step (Trainer.train(..., group)); BatchNorm
statistics stay per rank.  Prints one JSON line on rank 0: supervised targets/s and block tets/s (all cells touched by the
4-hop blocks), summed over ranks, time = max over ranks."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.learning.runModel import Metrics, Trainer, adjust_learning_rate
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.sampler import NeighborSampler
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--updated", action="store_true", help="surfaceNetUpdatedEdgeFilters ('sage+', edge embeddings chained layer to layer)")
ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="bf16: activations / phi stored in bf16, bf16 MFMA, fp32 accumulate and master weights")
ap.add_argument("--points", type=int, default=150000)
ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--widths", type=str, default=None, help="conv widths, e.g. 128,256,512,1024 (configs/modelnet.yaml:56; batch 1024 there) or 64,128,256,512 "
                                                          "(eth / aerial / terrestrial); default: the shipped checkpoint's 64,128,128,128")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=300, help="untimed steps; the GPU needs ~0.5 s of this load before its step time settles (measured: 2.4 -> 1.5 ms over the first ~300 steps of a process)")
ap.add_argument("--prefetch", choices=["none", "stream", "thread"], default="stream",
                help="block builder: in line | one block ahead on a side stream, issued by the library's builder thread (default) | "
                     "ahead in a Python worker thread")
ap.add_argument("--fresh-blocks", action="store_true", help="fresh tensors for every block instead of the builder's ring of three buffer sets")
ap.add_argument("--no-roofline", action="store_true", help="skip the GEMM / aggregate replays behind the `roofline` object")
args = ap.parse_args()

rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
if world != args.gpus:
    raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
backend = os.environ.get("DGNN_BENCH_BACKEND", "nccl")
if backend != "nccl":
    local_rank %= max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local_rank)
dev = "cuda:%d" % local_rank
if world > 1:
    import torch.distributed as dist
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device(dev))
    else:
        dist.init_process_group(backend)

points, batch, steps = args.points, args.batch, args.steps
adj, _, _ = delaunay_tet_graph(points, rank)          # one scene shard per GPU: every rank its own scene
n = adj.shape[0] // 4
ei = torch.from_numpy(adj.T.astype(np.int64)).to(dev)
x = hashed_normal(np.arange(n), 29, seed=1 + 10 * rank, device=dev)
x[:, 0] = x[:, 0].abs() + 0.05
ea = hashed_normal(np.arange(4 * n), 20, seed=2 + 10 * rank, device=dev)
occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
all_ = Config(x=x, y=torch.cat([occ, 1 - occ], 1), edge_attr=ea)
widths = [int(v) for v in args.widths.split(",")] if args.widths else [64, 128, 128, 128]
clf = reconbench_pretrained(device=dev, convs=tuple(widths))
clf.temp.current_epoch = 0
clf.training.metrics = Metrics()
torch.manual_seed(0)
if args.updated:
    import torch.nn.functional as F
    from dgnn_amd.learning.surfaceNetUpdatedEdgeFilters import SurfaceNet as UpdatedNet
    from dgnn_amd.partition import allreduce_gradients
    uclf = Config.wrap(dict(training=dict(model_params=list(widths), model_name="sage+", loss="kl"),
                            features=dict(normalization_feature=1, keep_normalization_feature=0), temp=dict(device=dev)))
    net = UpdatedNet(28, uclf).to(dev).train()

    class _Tr:   # the reference's loss (runModel.py:171-211) on the Updated model's (x, edge_attr, n_id, adjs) batch layout
        def train(self, data, opt, clf, group=None):
            from dgnn_amd import functional as Fn
            d_u = Config(x=data.all.x, edge_attr=data.all.edge_attr, n_id=data.batch_n_id, adjs=data.batch_adjs)
            if os.environ.get("DGNN_TRAIN_DIRECT", "1") != "0":
                # the step without the autograd engine (round 6: SurfaceNet.train_step_direct of the Updated model; same kernels, same numbers)
                bx, by = Trainer._batch_rows(data, data.batch_adjs[-1][2][1])

                def loss_fn(logits):
                    got = Fn.ops.kl_cell_loss_step(logits, by, bx[:, 0], 0)
                    if got is None:
                        loss_, sums_ = Fn.ops.kl_cell_loss_fwd(logits, by, bx[:, 0], 0)
                        return loss_, Fn.ops.kl_cell_loss_bwd(logits, by, bx[:, 0], 0, sums_, torch.ones((), device=logits.device))
                    return got[0], got[2]
                loss = net.train_step_direct(d_u, loss_fn)
                if loss is not None:
                    allreduce_gradients(net, group)
                    opt.step()
                    return loss
            opt.zero_grad()
            logits = net(d_u).float()
            bx, by = Trainer._batch_rows(data, data.batch_adjs[-1][2][1])            # x[ids], y[ids]: from the block builder when it gathered them
            loss, _ = Fn.kl_cell_loss(logits, by, bx[:, 0])   # the Trainer's fused loss (runModel.py:171-209)
            loss.backward()
            allreduce_gradients(net, group)
            opt.step()
            return loss

        attach_block_rows = staticmethod(Trainer.attach_block_rows)
    tr = _Tr()
else:
    net = SurfaceNet(clf).to(dev).train()
    tr = Trainer(net)
if args.dtype == "bf16":
    net.set_storage_dtype(torch.bfloat16)
if world > 1:
    from dgnn_amd.partition import broadcast_parameters
    broadcast_parameters(net)
if os.environ.get("DGNN_TORCH_ADAM") == "1":
    opt = torch.optim.Adam(net.parameters(), lr=clf.training.learning_rate, fused=True)
else:
    from dgnn_amd.learning.runModel import make_adam
    opt = make_adam(net.parameters(), clf.training.learning_rate)      # what Trainer.train_test builds: one library launch per step
adjust_learning_rate(opt, clf)
g = torch.Generator().manual_seed(rank)
per = (n // batch) * batch     # whole batches per permutation: no duplicate targets inside a batch
need = batch * (steps + args.warmup)
idx = torch.cat([torch.randperm(n, generator=g)[:per] for _ in range(need // per + 1)])[:need]
loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=idx.to(dev), num_nodes=n, batch_size=batch,
                         prefetch={"none": False, "stream": True, "thread": "thread"}[args.prefetch], reuse_buffers=not args.fresh_blocks)
if os.environ.get("DGNN_BLOCK_ROWS", "1") != "0":
    tr.attach_block_rows(loader, all_, net)      # x[n_id, 1:], x[ids], y[ids] gathered by the block builder behind every block (as Trainer.train_test does)
it = iter(loader)
block = 0
for _ in range(args.warmup):
    bs, n_id, adjs = next(it)
    tr.train(Config(all=all_, batch_n_id=n_id, batch_adjs=adjs), opt, clf)


def sync():
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


sync()
t0 = time.perf_counter()
for _ in range(steps):
    bs, n_id, adjs = next(it)
    block += int(n_id.numel())
    loss = tr.train(Config(all=all_, batch_n_id=n_id, batch_adjs=adjs), opt, clf)
host_dt = time.perf_counter() - t0      # the main thread has issued every step; what is left until sync() returns is the GPU's backlog
sync()
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt, float(block)], device=dev, dtype=torch.float64)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dt, block = float(tmax[0]), float(t[1])


def replay_roofline():
    """The step's dominant kernel by GPU time is the fp32-class GEMM (k_linear_fwd_x3, 15 launches per step: conv / decoder GEMMs
    forward, input gradients backward).  Its launches are replayed at the LAST block's shapes (HIP events on the current stream,
    10 launches each): achieved = their algorithmic flops / their time.  Peak: the dense bf16 MFMA rate / 6 (every fp32 product
    is 6 bf16 partial products).  The HBM-bound aggregate backward (k_agg_bwd_c, the longest launches) is reported next to it."""
    from dgnn_amd import ops
    from dgnn_amd.graph import plan_for

    def timed(f, it=10):
        f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it

    w = [28] + list(widths)
    if args.updated:
        # Updated variant: the longest launches are the given-phi aggregate backward (k_agg_bwd_g) of the outer blocks: reads x, da and the edge
        # embeddings phi [E, C] once, writes dx and dphi [E, C]
        e, e_id, size = adjs[1]
        c = w[1]
        plan = plan_for(e, size[0], size[1])
        tp = plan.transposed
        dt_ = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        esz = 2 if args.dtype == "bf16" else 4
        xx, da = torch.randn(size[0], c, device=dev).to(dt_), torch.randn(size[1], c, device=dev).to(dt_)
        phi = torch.randn(e.size(1), c, device=dev).to(dt_)
        t_b = timed(lambda: ops.aggregate_bwd(tp[0], tp[1], tp[2], size[0], plan.rowptr, xx, da, phi=phi))
        bytes_b = ((2 * size[0] + size[1]) * c + 2 * e.size(1) * c) * esz + e.size(1) * 8
        return {"bound": "hbm", "kernel": "k_agg_bwd_g (given-phi aggregate backward of the %d-channel layer, lane-group form: reads x, da, phi [E, C]; writes dx, dphi [E, C]; "
                                           "the longest launches of the step are its instances)" % c, "achieved": round(bytes_b / t_b / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(bytes_b / t_b / 1e6 / 8000.0, 4), "traffic": None, "algorithmic_bytes_per_launch": bytes_b, "avg_launch_ms": round(t_b, 4),
                "timing": "the launch replayed 10x at the last block's shapes, HIP events on the launching stream"}
    gemms, flops, ms = [], 0.0, 0.0
    for i, (e, e_id, size) in enumerate(adjs):
        n_src, n_dst = size
        gemms.append((n_dst, w[i], w[i], w[i + 1]))                      # z = a.Wj^T + x_dst.Wi^T
        gemms += [(n_dst, w[i + 1], 0, w[i])] * (2 if i > 0 else 1)      # da = dz.Wj (+ dx_dst += dz.Wi)
    hd = w[-1] // 2
    gemms += [(batch, w[-1], 0, hd), (batch, hd, 0, 2), (batch, 2, 0, hd), (batch, hd, 0, w[-1])]   # decoder forward / backward
    for M, k1, k2, no in gemms:
        A1, W1 = torch.randn(M, k1, device=dev), torch.randn(no, k1, device=dev)
        A2, W2 = (torch.randn(M, k2, device=dev), torch.randn(no, k2, device=dev)) if k2 else (None, None)
        out = torch.empty(M, no, device=dev)
        ms += timed(lambda: ops.linear_fwd(A1, W1, A2, W2, out=out))
        flops += 2.0 * M * (k1 + k2) * no
    e, e_id, size = adjs[1]                     # second conv layer (64 channels at the shipped widths): the largest launch that also writes dx
    c1 = w[1]
    plan = plan_for(e, size[0], size[1])
    tp, rows = plan.transposed, plan.transposed_edge_rows
    x, da = torch.randn(size[0], c1, device=dev), torch.randn(size[1], c1, device=dev)
    We, be = torch.randn(c1, 20, device=dev), torch.randn(c1, device=dev)
    t_b = timed(lambda: ops.aggregate_bwd(tp[0], tp[1], rows if rows is not None else tp[2], size[0], plan.rowptr, x, da, all_.edge_attr if rows is not None
                                          else all_.edge_attr[e_id], We, be))
    bytes_b = size[0] * c1 * 4 * 2 + e.size(1) * 88 + size[1] * c1 * 4 + size[0] * 4
    peak = 2500.0 / 6
    # primary object: the HBM-bound gather / scatter kernel with the longest single launch; the GEMMs (largest share of the
    # step's GPU time, but 15 small launches: M <= 70k rows, K <= 256) are reported next to it
    return {"bound": "hbm", "kernel": "k_agg_bwd_c<1,20> (aggregate backward of the %d-channel layer, chunked form: gathers da rows, recomputes the filter, "
                                       "writes dx, dWe, dbe; the longest launches of the step are its instances)" % c1, "achieved": round(bytes_b / t_b / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(bytes_b / t_b / 1e6 / 8000.0, 4), "traffic": None, "algorithmic_bytes_per_launch": bytes_b, "avg_launch_ms": round(t_b, 4),
            "timing": "each launch replayed 10x at the last block's shapes, HIP events on the launching stream",
            "gemm": {"bound": "mfma", "kernel": "k_linear_fwd_x3 (%d launches per step: conv / decoder GEMMs forward, input gradients backward)" % len(gemms),
                     "achieved": round(flops / ms / 1e9, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(flops / ms / 1e9 / peak, 4),
                     "algorithmic_flops_per_step": flops, "gemm_ms_per_step": round(ms, 4),
                     "peak_note": "dense bf16 MFMA 2.5 PFLOP/s / 6 partial products per fp32-class product; launches of 2k..70k rows are latency-bound"}}


replicas = None
if world > 1:
    # data-parallel replicas must hold the same weights after every all-reduced step: one fp64 checksum per rank, compared on rank 0
    cs = torch.stack([p.detach().double().abs().sum() for p in net.parameters()]).sum().reshape(1).cpu()
    sums = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, cs, group=dist.new_group(backend="gloo"))
    replicas = {"param_abs_sum_per_rank": [float(v) for v in sums], "equal": all(float(v) == float(sums[0]) for v in sums)}
roof = None
if rank == 0 and not args.no_roofline and (args.updated or args.dtype == "f32"):
    roof = replay_roofline()
if rank == 0:
    print(json.dumps({"metric": "training step (block builder + fwd + bwd + %sAdam), %d x MI355X" % ("gradient all-reduce + " if world > 1 else "", world),
                      "model": "UpdatedEdgeFilters sage+" if args.updated else "StaticEdgeFilters", "dtype": args.dtype, "n_gpus": world,
                      "parallelism": "data-parallel replicas, one scene shard per GPU, flat RCCL all-reduce" if world > 1 else "single GPU",
                      "targets_per_s": round(batch * steps * world / dt, 1), "block_tets_per_s": round(block / dt, 1),
                      "ms_per_step": round(dt / steps * 1e3, 3), "host_issue_ms_per_step": round(host_dt / steps * 1e3, 3), "batch_targets_per_gpu": batch,
                      "avg_block_tets": round(block / steps / world, 1), "steps": steps, "block_builder": args.prefetch, "scene_tets_per_gpu": n, "final_loss": float(loss), "replicas": replicas, "roofline": roof}))
if world > 1:
    dist.destroy_process_group()
