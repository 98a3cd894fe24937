#!/usr/bin/env python3
"""Benchmark of the dgnn hot path on MI355X: whole-graph in/out classification of a synthetic Delaunay
tetrahedron graph (SurfaceNet.inference_layer equivalent: 4 x [edge-filtered SAGE conv + BN(eval) +
ReLU] + decoder -> logits [N,2]), fp32, shipped kf96 weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over the graph, inputs resident in HBM in the reference's layout
(x [N,29] fp32, edge_attr [4N,20] fp32, edge_index [2,4N] int64).  The step INCLUDES building the
graph plan (stable destination sort) and staging edge_attr into plan order, because the reference
takes a raw edge_index on every call.  N>1: the scene has gpus x 150k points and is partitioned
spatially, one part per rank, with an RCCL halo exchange of boundary-tet features before conv
layers 1..3 (weak scaling: ~1M tets per GPU).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_TET = 5048          # algorithmic HBM bytes per tet, whole path, fp32
LAYER_BYTES = {(28, 64): 704, (64, 128): 1104, (128, 128): 1360}  # per tet, per fused layer launch
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def load_weights():
    w = np.load(os.path.join(ROOT, "tests", "golden", "kf96_weights.npz"))
    return {k: torch.from_numpy(w[k]) for k in w.files}


def make_scene(points, seed):
    from dgnn_amd.synthetic import delaunay_tet_graph
    adj, cent, _ = delaunay_tet_graph(points, seed)
    n = adj.shape[0] // 4
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 29, generator=g)
    ea = torch.randn(4 * n, 20, generator=g)
    return adj, cent, x, ea


def cpu_baseline(points, threads):
    """Times the CPU oracle (plain-PyTorch restatement of the reference path) on a bounded sample of the
    same workload: same generator, `points` points (~6.7 tets per point)."""
    from dgnn_amd.config import Config, reconbench_pretrained
    from oracle.static_edge_filters import SurfaceNet as OracleNet
    adj, _, x, ea = make_scene(points, 0)
    n = adj.shape[0] // 4
    net = OracleNet(reconbench_pretrained(device="cpu"))
    net.load_state_dict(load_weights())
    net.eval()
    data = Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(adj.T.astype(np.int64)))
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    times = []
    with torch.no_grad():
        net.inference_layer(data)
        for _ in range(3):
            t0 = time.perf_counter()
            net.inference_layer(data)
            times.append(time.perf_counter() - t0)
    torch.set_num_threads(old)
    return n / float(np.median(times)), n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=150000, help="Delaunay points per GPU (150000 -> 1 010 078 tets)")
    ap.add_argument("--cpu-points", type=int, default=30000, help="sample size of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cached-plan", action="store_true", help="reuse the graph plan across steps (reported, not the metric)")
    ap.add_argument("--gemm-mode", choices=["f32", "bf16x3", "bf16x3f"], default=None,
                    help="dense part of the fused layer: exact fp32 MFMA, or 3-way split-bf16 MFMA (fp32-class accuracy)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # DGNN_BENCH_BACKEND=gloo: validation runs of the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # devices, halo rows are staged through host memory).  The numbers of such a run are not a benchmark.
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

    from dgnn_amd import ops
    from dgnn_amd.config import Config, reconbench_pretrained
    from dgnn_amd.graph import GraphPlan
    from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet

    if args.gemm_mode is not None:
        ops.GEMM_MODE = {"f32": ops.GEMM_F32, "bf16x3": ops.GEMM_BF16X3, "bf16x3f": ops.GEMM_BF16X3_FILTER}[args.gemm_mode]
    net = SurfaceNet(reconbench_pretrained(device=dev))
    net.load_state_dict(load_weights())
    net = net.to(dev).eval()

    if world == 1:
        adj, _, x, ea = make_scene(args.points, 0)
        n_total = n_local = adj.shape[0] // 4
        data = Config(x=x.to(dev), edge_attr=ea.to(dev), edge_index=torch.from_numpy(adj.T.astype(np.int64)).to(dev))
        cached = {}

        def step():
            plan = cached.get("p")
            if plan is None:
                plan = GraphPlan(data.edge_index, n_local, n_local, hint=ops.PLAN_HINT_REFERENCE)
                if args.cached_plan:
                    cached["p"] = plan
            return net.inference_layer(data, plan=plan)
        workload = "synthetic Delaunay tet graph, %d points -> N=%d tets, E=%d, whole-graph inference_layer" % (args.points, n_total, 4 * n_total)
    else:
        from dgnn_amd.partition import PartitionedScene
        scene = PartitionedScene.build_synthetic(args.points * world, 0, rank, world, dev)
        n_total, n_local = scene.n_total, scene.n_own

        def step():
            return scene.inference_layer(net)
        transport = "RCCL" if backend == "nccl" else "host-staged %s (validation run, not a benchmark)" % backend
        if backend == "nccl":
            # Safety net: if the device-to-device exchange cannot run on this node (P2P/IPC disabled ...), every rank sees the
            # error in its first step; all ranks then agree to stage the halo rows through host memory over gloo, and the
            # JSON line says so.  Compute is unchanged.
            import torch.distributed as dist
            from dgnn_amd.partition import HaloExchange
            gloo = dist.new_group(backend="gloo")
            failed = 0
            try:
                step()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                failed = 1
                sys.stderr.write("rank %d: RCCL halo exchange failed (%s)\n" % (rank, e))
            flag = torch.tensor([failed])
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=gloo)
            if int(flag.item()):
                scene.exchange = HaloExchange(scene.lp, dev, pack=ops.gather_rows, group=gloo, via_host=True)
                transport = "host-staged gloo (RCCL point-to-point failed on this node)"
        workload = "synthetic Delaunay scene, %d points -> N=%d tets, %d-way spatial partition + %s halo exchange overlapped with interior cells" % (
            args.points * world, n_total, world, transport)

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # HIP events around every launch of the dominant kernel (fused 128->128 layer) INSIDE the timed region, recorded on
    # the stream the kernel is launched on (torch's current stream)
    dom_events = []

    def hook(tok, c_in, c_out, n_dst):
        if (c_in, c_out) != (128, 128):
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if tok is None:
            return ev
        dom_events.append((tok, ev, n_dst))
        return None
    ops.FUSED_LAUNCH_HOOK = hook
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    ops.FUSED_LAUNCH_HOOK = None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = n_total * args.steps / dt

    # ---- roofline of the dominant kernel (fused 128->128 layer), HIP events on the launch stream ----
    roof = None
    breakdown = {}
    if world == 1:
        plan = GraphPlan(data.edge_index, n_local, n_local, hint=ops.PLAN_HINT_REFERENCE)
        xs = data.x[:, 1:]
        in_kernel = ops.EDGE_GATHER_IN_KERNEL  # fused layers gather edge rows by eid themselves: no staging pass
        ea_l, eid_l = (data.edge_attr, plan.eid) if in_kernel else (plan.sorted_edge_attr(data.edge_attr), None)
        # per-kernel timing: replay each layer 10x between events
        def timed(fn, reps=10):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        breakdown["plan_ms"] = timed(lambda: GraphPlan(data.edge_index, n_local, n_local, hint=ops.PLAN_HINT_REFERENCE), 5)
        breakdown["edge_sort_ms"] = 0.0 if in_kernel else timed(lambda: ops.gather_rows(data.edge_attr, plan.eid), 5)
        h = xs
        for i in range(4):
            conv = net.convs[i][0]
            scale, shift = net._fold(net.convs[i][1], conv.lin_j.out_features, dev)
            hin = h
            if ops.fused_layer_supported(hin.size(1), conv.lin_j.out_features, 20):
                fn = lambda hin=hin, conv=conv, scale=scale, shift=shift: ops.sage_layer_fused_fwd(
                    plan.rowptr, plan.src, n_local, hin, ea_l, conv.lin_e.weight, conv.lin_e.bias, conv.lin_j.weight,
                    conv.lin_j.bias, conv.lin_i.weight, scale, shift, True, eid=eid_l)
            else:
                fn = lambda hin=hin, i=i: net._eval_layers_one(i, hin, data.edge_attr, plan)
            breakdown["layer%d_ms" % i] = timed(fn)
            h = fn()
        breakdown["decoder_ms"] = timed(lambda: net._eval_decoder(h))
    dom_name = {0: "k_sage_fused<128,128,0>", 1: "k_sage_fused<128,128,1>", 2: "k_sage_fused_mfma<128,128>"}[ops.GEMM_MODE]
    if dom_events:
        # average launch of the dominant kernel over the timed region (layers 2 and 3 of every step; with a partition the
        # interior and boundary launches of a layer are added up so that bytes and time cover the same rows)
        tot_ms = sum(a.elapsed_time(b) for a, b, _ in dom_events)
        tot_rows = sum(r for _, _, r in dom_events)
        launches_per_layer = len(dom_events) / (2.0 * args.steps)
        dom_ms = tot_ms / len(dom_events) * launches_per_layer
        algo = int(LAYER_BYTES[(128, 128)] * tot_rows / (2.0 * args.steps))
        achieved = algo / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the
        # gfx950 note + WRITE_SIZE); scaled by tets when the bench graph differs from the profiled one
        traffic, pmc = None, {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01f_final_traffic.json")))
            traffic = round(tj["traffic_bytes_per_launch"] * n_local / 1010078)
            pmc = {k: tj[k] for k in ("mfma_busy_frac", "valu_busy_frac", "tcc_hit_rate", "clock_ghz") if k in tj}
        except Exception:
            pass
        roof = {"bound": "hbm", "kernel": dom_name + " (layers 2 and 3)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(dom_ms, 4),
                "timing": "HIP events around each launch inside the timed steps (%d launches)" % len(dom_events),
                "pmc": dict(pmc, source="profiles/r01f_final.md (rocprofv3 --pmc passes of this command)"),
                "whole_path_frac": round(value * BYTES_PER_TET / 1e9 / HBM_PEAK_GBS / world, 4)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # 16 threads is where this op chain peaks on the GPU host (tools/cpu_sweep.py: 1 -> 9.7e4, 16 -> 1.6e5,
        # 64 -> 1.1e5, 128 -> 6.7e4 tets/s on a 256-core box): more threads only add contention
        cores = min(os.cpu_count() or 1, 16)
        v, n_s = cpu_baseline(args.cpu_points, cores)
        v1, _ = cpu_baseline(args.cpu_points, 1)
        cpu = {"value": round(v, 1), "unit": "tets/s", "cores": cores, "single_thread_value": round(v1, 1), "kind": "port",
               "sample": "oracle (PyTorch-CPU restatement of inference_layer), same generator at %d points -> %d tets, "
                         "1 warm-up + median of 3" % (args.cpu_points, n_s)}

    if rank == 0:
        out = {
            "metric": "tetrahedra/sec (in/out classified), 1M-tet graph at 1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "tets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "tets_per_gpu": n_local, "weights": "kf96 checkpoint [64,128,128,128]",
                       "plan_in_step": not args.cached_plan,
                       "gemm": {0: "fp32 MFMA", 1: "split-bf16 MFMA for the dense part (3 exact bf16 parts per fp32 operand, 6 products, fp32 accumulate)",
                                2: "split-bf16 MFMA for the dense part and the filter MLP (3 exact bf16 parts per fp32 operand, 6 products, fp32 accumulate)"}[ops.GEMM_MODE], "breakdown_ms": {k: round(v, 4) for k, v in breakdown.items()}},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
