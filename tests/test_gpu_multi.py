"""The multi-rank entry points as the driver launches them -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` -- run as
fresh child processes on the test box's single GPU (DGNN_BENCH_BACKEND=gloo: the ranks share the device, halo rows / gradients are staged through
host memory; the stream choreography, the partition, the plan builder and every kernel are the product's), and BASELINE config 4 at its size:
the 10 026 136-tet scene cut 8 ways, ranks emulated in one process, against the whole-graph run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from dgnn_amd.config import Config
from test_gpu_parity import DEV, hip_static

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def torchrun(script, args, nproc=2, timeout=900, extra_env=None):
    """fresh child processes (never a re-exec of this one); bench.py prints a <= 4 KB headline and writes the full nested object to
    gpurun_out/bench_full_<DGNN_BENCH_TAG>.json -- the full object is what comes back (its headline under "_headline")"""
    tag = "t%d_%d" % (os.getpid(), _free_port())
    env = dict(os.environ, DGNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", DGNN_BENCH_TAG=tag)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert len(lines[0]) <= 4096, len(lines[0])
    full = os.path.join(ROOT, "gpurun_out", "bench_full_%s.json" % tag)
    if "full" in d and os.path.exists(full):
        with open(full) as f:
            whole = json.load(f)
        os.remove(full)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling"):      # the headline is the full object's own numbers
            assert d.get(k) == whole.get(k), k
        whole["_headline"] = d
        whole["_stderr"] = r.stderr[-6000:]
        return whole
    d["_stderr"] = r.stderr[-6000:]
    return d


@pytest.mark.parametrize("scaling,halo", [("strong", "recompute"), ("weak", "recompute"), ("strong", "exchange")])
def test_bench_py_two_ranks_as_the_driver_launches_it(scaling, halo):
    """bench.py --gpus 2: ONE JSON line from rank 0, the contract's keys, and a `check` at N > 1: the logits of both ranks' cells, gathered on rank 0,
    are bit-identical to the same scene run as one whole graph by a single rank and agree with the CPU oracle; the other scaling mode is nested
    with its own check."""
    d = torchrun("bench.py", ["--gpus", "2", "--scaling", scaling, "--steps", "2", "--warmup", "1", "--no-train", "--points", "30000"] +
                 (["--halo", halo] if halo != "recompute" else []))       # (recompute = rings of halo cells, no exchange: the default)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "check", "other_scaling"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == scaling and d["unit"] == "tets/s" and d["vs_baseline"] is None
    n_one = 201062                                     # 30000 points
    n_main = d["check"]["n_tets"]
    assert (n_main == n_one) == (scaling == "strong") and abs(d["value"] - n_main / d["ms_per_step"] * 1e3) <= 1e-3 * d["value"]
    c = d["check"]
    assert c["ok"] and c["bit_identical_to_single_rank"] and c["cells_covered"] == n_main and c["max_abs_diff_vs_single_rank"] == 0.0
    assert c["vs_cpu_oracle"]["ok"] and c["vs_cpu_oracle"]["max_abs_err"] <= 1e-4
    o = d["other_scaling"]
    assert o["scaling"] == ("weak" if scaling == "strong" else "strong") and o["check"]["ok"] and o["check"]["bit_identical_to_single_rank"]
    # says what it was: a validation run, not a benchmark
    assert ("host-staged gloo" if halo == "exchange" else "validation run under gloo") in d["config"]["workload"]
    assert ("no collective in the data path" in d["config"]["workload"]) == (halo == "recompute")
    assert 0.2 < d["config"]["tets_per_gpu"] / (n_main / 2) < 1.8
    # round 5: north_star's exchange form rides in every ring-parts line (here host-staged: the ranks share one GPU under gloo)
    assert ("halo_exchange" in d) == (halo == "recompute")
    if halo == "recompute":
        h = d["halo_exchange"]
        for k in ("ms_per_step", "value", "transport", "rccl_world", "check", "n_tets", "halo_rows_this_rank"):
            assert k in h, k
        assert h["n_tets"] == n_main and h["check"]["ok"] and h["check"]["bit_identical_to_single_rank"] and h["rccl_world"] is None and "host-staged" in h["transport"]
        assert h["halo_rows_this_rank"] > 0 and abs(h["value"] - n_main / h["ms_per_step"] * 1e3) <= 1e-3 * h["value"]


@pytest.mark.parametrize("extra", [["--updated"], ["--updated", "--dtype", "bf16"], []])
def test_bench_train_two_ranks_data_parallel(extra):
    """BASELINE config 5's shape of work at world 2 (UpdatedEdgeFilters, one scene shard per rank, one flat all-reduce per step; also the Static
    model): the launch line of the driver, replicas equal after the all-reduced steps, finite loss"""
    d = torchrun(os.path.join("tools", "bench_train.py"), ["--gpus", "2", "--points", "12000", "--batch", "256", "--steps", "4", "--warmup", "3", "--no-roofline"] + extra)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and np.isfinite(d["final_loss"]) and d["final_loss"] > 0
    assert d["replicas"]["equal"], d["replicas"]
    assert ("Updated" in d["model"]) == ("--updated" in extra)


def test_config4_10m_tets_eight_parts_equal_the_whole_graph():
    """BASELINE config 4 at its size on ONE GPU: the 1 485 000-point scene (10 026 136 tets) resident in HBM, rcb_partition(.., 8), every part's
    layers run as the partitioned forward runs them (interior cells, then boundary cells, halo rows delivered as the exchange delivers them) --
    the union of the 8 parts' logits must equal the whole-graph inference_layer BIT FOR BIT; halo sizes against SURVEY 8e's estimate."""
    from dgnn_amd.graph import GraphPlan
    from dgnn_amd.partition import build_local_part, rcb_partition
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    world = 8
    adj, cent, _ = delaunay_tet_graph(1485000, 0)
    n = adj.shape[0] // 4
    assert n == 10026136
    ei = np.empty((2, 4 * n), np.int64)
    ei[0] = np.repeat(np.arange(n, dtype=np.int64), 4)
    ei[1] = adj[:, 1]
    del adj
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    net = hip_static()
    ei_d = torch.from_numpy(ei).to(DEV)
    full = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=ei_d))
    del ei_d
    part = rcb_partition(cent, world)
    lps = [build_local_part(ei, part, r, world) for r in range(world)]
    del ei
    # ~1.25M cells per rank, halo ~1.7 % of the owned rows (3.65 % at 1M tets scaled by 10^(-1/3)), every cell owned exactly once
    assert sum(lp.n_own for lp in lps) == n
    for lp in lps:
        assert abs(lp.n_own - n / world) <= 2 and 0.008 <= lp.n_halo / lp.n_own <= 0.03, (lp.n_own, lp.n_halo)
        assert sum(lp.send_counts) > 0 and lp.n_interior / lp.n_own > 0.93
    print("halo rows per rank:", [lp.n_halo for lp in lps], "= %.2f %% of the owned rows" % (100 * np.mean([lp.n_halo / lp.n_own for lp in lps])))
    plans = [GraphPlan(torch.from_numpy(lp.edge_index).to(DEV), lp.n_own + lp.n_halo, lp.n_own, hint=1) for lp in lps]
    eas = [ea[torch.from_numpy(lp.edge_gid).to(DEV)] for lp in lps]     # a rank's edge rows, already in plan order
    del ea
    hs = [x[torch.from_numpy(np.concatenate([lp.own_gid, lp.halo_gid])).to(DEV)][:, 1:] for lp in lps]
    del x
    widths = [64, 128, 128, 128]
    own = [torch.from_numpy(lp.own_gid).to(DEV) for lp in lps]
    halo = [torch.from_numpy(lp.halo_gid).to(DEV) for lp in lps]
    fuse = net.fuses_decoder(3)        # the last layer's launches then carry the decoder and write logits (as PartitionedScene.inference_layer runs them)
    for i in range(net.num_layers):
        bufs = []
        for r, lp in enumerate(lps):
            dec = fuse and i == 3
            buf = torch.full((lp.n_own, 2) if dec else (lp.n_own + lp.n_halo, widths[i]), float("nan"), device=DEV)
            for b, e in ((0, lp.n_interior), (lp.n_interior, lp.n_own)):
                net._eval_layers(hs[r], lp.n_own, eas[r], [plans[r]] * 4, False, only=i, out=buf, rows=(b, e), decode=dec)
            bufs.append(buf)
        if dec:
            hs = bufs
            break
        glob = torch.empty(n, widths[i], device=DEV)
        for r in range(world):
            glob[own[r]] = bufs[r][:lps[r].n_own]
        for r in range(world):           # what the exchange delivers
            bufs[r][lps[r].n_own:] = glob[halo[r]]
        del glob
        hs = bufs
    logits = torch.full((n, 2), float("nan"), device=DEV)
    for r in range(world):
        logits[own[r]] = hs[r] if fuse else net._eval_decoder(hs[r][:lps[r].n_own])
    assert torch.equal(logits, full)


# ---- backward of a partitioned single scene with the HIP model (round 4; SURVEY 8e "Backward mirrors it") -------------------------------------------
def _ptrain_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dgnn_amd.partition import PartitionedScene, allreduce_gradients, build_local_part, partitioned_kl_loss, rcb_partition
    from test_gpu_parity import DEV, hip_static
    from test_partition_cpu import _scene
    adj, cent, x, ea = _scene(900, 6)
    x[:, 0] = x[:, 0].abs() + 0.05
    occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
    y = torch.cat([occ, 1 - occ], 1)
    lp = build_local_part(adj.T.astype(np.int64), rcb_partition(cent, world), rank, world)
    rows = torch.from_numpy(np.concatenate([lp.own_gid, lp.halo_gid]))
    own = torch.from_numpy(lp.own_gid)
    net = hip_static(train=True)
    scene = PartitionedScene(lp, x[rows], ea[torch.from_numpy(lp.edge_gid)], DEV)
    assert scene.exchange.via_host              # both ranks share the box's one GPU: rows and sums are staged through host memory under gloo
    logits = scene.train_forward(net)
    loss = partitioned_kl_loss(logits, y[own].to(DEV), x[own, 0].to(DEV), via_host=True)
    loss.backward()
    allreduce_gradients(net, average=False)
    torch.save(dict(gid=own, logits=logits.detach().cpu(), loss=loss.detach().cpu(), grads={k: p.grad.cpu() for k, p in net.named_parameters()},
                    buffers={k: b.cpu() for k, b in net.named_buffers()}), os.path.join(out_dir, "t%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_backward_hip_model_two_processes(tmp_path):
    """a scene cut across two processes, HIP model in training mode: halo gradients return to their owners (reverse exchange), BatchNorm statistics and
    their backward sums span both parts ([2 C] all-reduces around dgnn_bn_batch_stats / dgnn_bn_relu_bwd_sums / _apply), parameter gradients are summed --
    logits, loss, every parameter gradient and the running buffers against the fp64 oracle's single-process whole-graph step"""
    import socket
    import torch.multiprocessing as mp
    import torch.nn.functional as F
    from helpers import oracle_static
    from test_partition_cpu import _scene
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    mp.spawn(_ptrain_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    adj, cent, x, ea = _scene(900, 6)
    n = adj.shape[0] // 4
    x, ea = x.double(), ea.double()
    x[:, 0] = x[:, 0].abs() + 0.05
    occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
    y = torch.cat([occ, 1 - occ], 1)
    ei = torch.from_numpy(adj.T.astype(np.int64))
    onet = oracle_static(train=True, dtype=torch.float64)
    ologits = onet(Config(all=Config(x=x, edge_attr=ea), batch_n_id=torch.arange(n), batch_adjs=[(ei, torch.arange(4 * n), (n, n))] * 4))
    w = x[:, 0]
    oloss = (F.kl_div(F.log_softmax(ologits, dim=-1), y, reduction="none").sum(1) * w).sum() / w.sum()
    oloss.backward()
    outs = [torch.load(os.path.join(str(tmp_path), "t%d.pt" % r)) for r in range(2)]
    got = torch.full((n, 2), float("nan"), dtype=torch.float64)
    for o in outs:
        got[o["gid"]] = o["logits"].double()
    assert (got - ologits.detach()).abs().max().item() <= 2e-4 * max(1.0, ologits.abs().max().item())
    ograds = {k: p.grad for k, p in onet.named_parameters()}
    gmax = max(g.abs().max().item() for g in ograds.values())
    for o in outs:
        assert abs(o["loss"].item() - oloss.item()) <= 2e-5 * abs(oloss.item()) + 1e-9
        for k, g in o["grads"].items():
            err = (g.double() - ograds[k]).abs().max().item()
            assert err <= 5e-4 * ograds[k].abs().max().item() + 5e-6 * gmax, (k, err)
        for k, b in dict(onet.named_buffers()).items():
            assert (o["buffers"][k].double() - b.double()).abs().max().item() <= 1e-5 * max(b.double().abs().max().item(), 1e-30) + 1e-12, k


def test_bench_py_agrees_on_host_staging_when_the_first_exchange_fails():
    """bench.py --halo exchange, world 2: the first halo exchange fails on EVERY rank (fault injection, dgnn_amd/partition.py) -- the ranks agree
    (all-reduce(MAX) of the failure flags) to stage the halo rows through the host, say so in the line, and finish: rc 0, no hang (timeout), the same
    transport reported, logits bit-identical to the single-rank run.  VERDICT r5 item 8: this branch had never executed."""
    d = torchrun("bench.py", ["--gpus", "2", "--scaling", "strong", "--halo", "exchange", "--steps", "2", "--warmup", "1", "--no-train", "--points", "20000",
                              "--no-extras"], timeout=600, extra_env={"DGNN_FAULT_FIRST_EXCHANGE": "all", "DGNN_BENCH_SAFETY_NET": "1"})
    assert "RCCL point-to-point failed" in d["config"]["workload"], d["config"]["workload"]
    assert d["_stderr"].count("halo exchange failed") == 2                        # both ranks saw the failure and said so
    c = d["check"]
    assert c["ok"] and c["bit_identical_to_single_rank"] and c["cells_covered"] == c["n_tets"]


@pytest.mark.parametrize("fault", ["DGNN_FAULT_RCCL_UNAVAILABLE=1", "DGNN_FAULT_COMM_CREATE=all", "none"])
def test_native_halo_creation_failures_are_agreed_on(fault):
    """HaloExchange._init_native at world 2 with the library's communicator ATTEMPTED (DGNN_NATIVE_HALO=force: the agreement then runs over gloo on this
    one-GPU box): (a) RCCL reported unavailable on rank 1 ONLY -- all-reduce(MIN) keeps every rank out of the unique-id broadcast and ncclCommInitRank;
    (b) communicator creation failing on every rank -- the second agreement destroys what was made; (c) no injected fault -- on one GPU
    ncclCommInitRank itself refuses two ranks on one device, the same branch, reached the natural way.  Every time: both ranks end on the SAME transport
    (torch.distributed, host-staged), rc 0, no hang within the timeout, and the partitioned logits equal the single-rank run bit for bit."""
    env = {"DGNN_NATIVE_HALO": "force"}
    if fault != "none":
        k, v = fault.split("=")
        env[k] = v
    d = torchrun(os.path.join("tests", "halo_fault_child.py"), [], timeout=600, extra_env=env)
    assert d["world"] == 2 and d["ok"], d
    assert d["native_attempt"][0] != "ok" and d["native_attempt"][1] != "ok", d["native_attempt"]       # nobody got a communicator ...
    assert d["transport"][0] == d["transport"][1] == "torch.distributed", d["transport"]                 # ... and nobody was left on the other transport
    assert d["bit_identical_to_single_rank"] and d["cells_covered"] == d["n_tets"]
    if fault.startswith("DGNN_FAULT_RCCL_UNAVAILABLE"):
        assert all("unavailable on at least one rank" in a for a in d["native_attempt"]), d["native_attempt"]
    else:
        assert all("creation failed on at least one rank" in a for a in d["native_attempt"]), d["native_attempt"]
