"""Per-rank cost of the partitioned forward at N > 1, measured on ONE GPU: rank r's part of a W-way cut of the metric scene runs its real
launch chain (plan, layer 0, pack / interior / wait / boundary per later layer, decoder-carrying last launches) with the library's RCCL exchange --
the one process being its own peer: the rows the part would send to its peers come back as its halo rows (the halo rows are therefore not the
peers' rows and the logits of boundary cells are not the scene's -- this tool times; tests/test_gpu_infer.py, tests/test_gpu_multi.py and
bench.py's check at N > 1 verify).  `--transport loopback`: a device copy instead of RCCL (no process group).
Reports wall ms / step (host-issued) and the GPU-side ms / step from events, per rank, and the strong-scaling value they predict.

    python tools/bench_partition_rank.py --world 8 [--ranks 0,3] [--points 150000] [--steps 50] [--one-call 0|1]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", default="0")
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--one-call", type=int, default=-1, help="-1 = library default, 0 / 1 = per-layer Python chain / one library call per step")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "loopback"])
    ap.add_argument("--halo", default="recompute", choices=["recompute", "exchange"],
                    help="recompute: rings of halo cells resident, every layer recomputed on the rings later layers read, no exchange; exchange: one ring, RCCL between the layers")
    args = ap.parse_args()
    from dgnn_amd import ops, partition
    from dgnn_amd.config import reconbench_pretrained
    from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal, loader_cell_order
    dev = torch.device("cuda:0")
    if args.halo == "recompute":
        args.transport = "none"
    if args.transport == "rccl":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    net = SurfaceNet(reconbench_pretrained(device=dev, convs=(64, 128, 128, 128)))
    w = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kf96_weights.npz"))
    net.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files})
    net = net.to(dev).eval()
    if args.dtype == "bf16":
        net.set_storage_dtype(torch.bfloat16)
    adj, cent, _ = delaunay_tet_graph(args.points, 0)
    adj, cent, _ = loader_cell_order(adj, cent)
    part = partition.rcb_partition(cent, args.world)
    n = part.shape[0]
    ei = np.empty((2, 4 * n), dtype=np.int64)
    ei[0] = np.repeat(np.arange(n, dtype=np.int64), 4)
    ei[1] = adj[:, 1]
    out = []
    for r in [int(v) for v in args.ranks.split(",")]:
        lp = partition.build_ring_part(ei, part, r, args.world, net.num_layers) if args.halo == "recompute" else partition.build_local_part(ei, part, r, args.world)
        rows = np.concatenate([lp.own_gid, lp.halo_gid])
        n_send = int(sum(lp.send_counts))
        if args.transport == "rccl":      # its own peer: as many rows out as come in
            lp.send_idx = np.resize(lp.send_idx, lp.n_halo)
            lp.rank, lp.world, lp.send_counts, lp.recv_counts = 0, 1, [lp.n_halo], [lp.n_halo]
        sc = partition.PartitionedScene(lp, hashed_normal(rows, 29, seed=1, device=dev), hashed_normal(lp.edge_gid, 20, seed=2, device=dev), dev)
        if args.transport == "loopback":
            sc.exchange = partition.LoopbackExchange(lp, dev, pack=ops.gather_rows)
        elif args.transport == "rccl":
            assert sc.exchange._native is not None
        if args.one_call >= 0:
            sc.one_call = bool(args.one_call)
        for _ in range(args.warmup):
            sc.inference_layer(net)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            sc.inference_layer(net)
        e1.record()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        out.append(dict(rank=r, rings=lp.ring_counts, n_own=lp.n_own, n_interior=lp.n_interior, n_halo=lp.n_halo, n_send=n_send,
                        wall_ms=round(wall, 4), host_issue_ms=round(t_issue / args.steps * 1e3, 4), gpu_ms=round(e0.elapsed_time(e1) / args.steps, 4),
                        one_call=bool(getattr(sc, "used_one_call", False))))
    worst = max(o["wall_ms"] for o in out)
    print(json.dumps(dict(world=args.world, halo=args.halo, scene_tets=n, dtype=args.dtype, ranks=out, predicted_strong_value=round(n / worst * 1e3, 1),
                          transport=args.transport, note="the rank is its own peer: per-rank launch chain, exchange calls and host cost are the real ones, link time is not in it")))
    if args.transport == "rccl":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
