"""Child of tests/test_gpu_multi.py::test_native_halo_creation_failures_are_agreed_on, launched by torch.distributed.run at world 2 on ONE GPU under
gloo: a partitioned scene in the exchange form whose HaloExchange is told to attempt the library's RCCL communicator (DGNN_NATIVE_HALO=force) while a
fault is injected (csrc/halo.hip: DGNN_FAULT_*).  Prints one JSON line on rank 0: what every rank's attempt ended with, the transport every rank is on,
and whether the partitioned logits equal the single-rank whole-graph run bit for bit."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = "cuda:0"
dist.init_process_group("gloo")

from dgnn_amd.config import Config, reconbench_pretrained
from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
from dgnn_amd.partition import PartitionedScene
from dgnn_amd.synthetic import hashed_normal

w = np.load(os.path.join(ROOT, "tests", "golden", "kf96_weights.npz"))
net = SurfaceNet(reconbench_pretrained(device=dev))
net.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files})
net = net.to(dev).eval()

sc = PartitionedScene.build_synthetic(6000, 0, rank, world, dev, keep_global=True, halo="exchange", hops=net.num_layers)
ex = sc.exchange
mine_state = (ex.native_attempt if ex.native_attempt is not None else "not attempted", "library RCCL" if ex._native is not None else "torch.distributed")
logits = sc.inference_layer(net).float().cpu()
torch.cuda.synchronize()
parts = [None] * world if rank == 0 else None
dist.gather_object((torch.from_numpy(sc.lp.own_gid), logits, mine_state), parts, dst=0)
if rank == 0:
    n = sc.n_total
    got = torch.full((n, 2), float("nan"))
    for gid, lg, _ in parts:
        got[gid] = lg
    ei = torch.empty((2, 4 * n), dtype=torch.int64)
    ei[0] = torch.arange(n).repeat_interleave(4)
    ei[1] = torch.from_numpy(sc.global_dst.astype(np.int64))
    xw, eaw = hashed_normal(np.arange(n), 29, seed=1, device=dev), hashed_normal(np.arange(4 * n), 20, seed=2, device=dev)
    whole = net.inference_layer(Config(x=xw, edge_attr=eaw, edge_index=ei.to(dev))).float().cpu()
    same = bool(torch.equal(got, whole))
    covered = int(torch.isfinite(got).all(1).sum())
    print(json.dumps({"world": world, "n_tets": n, "native_attempt": [p[2][0] for p in parts], "transport": [p[2][1] for p in parts],
                      "bit_identical_to_single_rank": same, "cells_covered": covered, "ok": same and covered == n}))
dist.barrier()
dist.destroy_process_group()
