"""Whole-scene inference as ONE library call (dgnn_static_infer_fwd, csrc/infer.hip): the call issues the launches the
per-layer entry points issue, so its logits must be BIT-IDENTICAL to the layer-by-layer path -- which the other suites hold to the oracle and
to the reference's golden logits -- on every graph shape, plan source and arithmetic mode; plus the oracle directly on one scene."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from dgnn_amd.config import Config
from helpers import gold, oracle_static
from test_gpu_parity import DEV, TOL_LOGIT, hip_static

pytestmark = pytest.mark.gpu


def _scene(points, seed=0, layout="transposed"):
    from dgnn_amd.synthetic import delaunay_tet_graph
    adj, _, _ = delaunay_tet_graph(points, seed=seed)
    n = adj.shape[0] // 4
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 29, generator=g)
    ea = torch.randn(4 * n, 20, generator=g)
    if layout == "transposed":      # the reference's view of its [E,2] array (processing/data.py:437-438)
        ei = torch.from_numpy(adj.astype(np.int64)).to(DEV).t()
    else:
        ei = torch.from_numpy(np.ascontiguousarray(adj.T.astype(np.int64))).to(DEV)
    return n, x, ea, ei


def _both_paths(net, data, plan=None, fresh=True):
    from dgnn_amd import ops
    from dgnn_amd.graph import clear_plan_cache
    assert ops.INFER_ONE_CALL
    if fresh:
        clear_plan_cache(data.edge_index)
    one = net.inference_layer(data, plan=plan)
    ops.INFER_ONE_CALL = False
    try:
        if fresh:
            clear_plan_cache(data.edge_index)
        ref = net.inference_layer(data, plan=plan)
    finally:
        ops.INFER_ONE_CALL = True
    return one, ref


@pytest.mark.parametrize("points,layout", [(5, "transposed"), (9, "contiguous"), (40, "transposed"), (600, "transposed"), (600, "contiguous"), (9000, "transposed")])
def test_one_call_equals_the_per_layer_path(points, layout):
    n, x, ea, ei = _scene(points, seed=points, layout=layout)
    net = hip_static()
    data = Config(x=x.to(DEV), edge_attr=ea.to(DEV), edge_index=ei)
    one, ref = _both_paths(net, data)
    assert one.shape == (n, 2) and torch.equal(one, ref)
    # the plan the call built hangs off edge_index like plan_for's: the next call reuses it (and is the same again)
    held = getattr(ei, "_dgnn_plans", None)
    assert held and len(held) == 1
    again, _ = _both_paths(net, data, fresh=False)
    assert torch.equal(again, ref)
    plan = next(iter(held.values()))
    want = np.argsort(ei[1].cpu().numpy(), kind="stable")
    assert np.array_equal(plan.eid.cpu().numpy(), want) and np.array_equal(plan.src.cpu().numpy(), ei[0].cpu().numpy()[want])


def test_one_call_vs_oracle_and_with_a_given_plan():
    from dgnn_amd import ops
    from dgnn_amd.graph import GraphPlan
    n, x, ea, ei = _scene(1500, seed=3)
    net = hip_static()
    data = Config(x=x.to(DEV), edge_attr=ea.to(DEV), edge_index=ei)
    one = net.inference_layer(data)
    with torch.no_grad():
        want = oracle_static().inference_layer(Config(x=x, edge_attr=ea, edge_index=ei.cpu()))
    assert (one.cpu() - want).abs().max().item() <= TOL_LOGIT
    for hint in (ops.PLAN_HINT_REFERENCE, ops.PLAN_HINT_GENERIC):
        got, ref = _both_paths(net, data, plan=GraphPlan(ei, n, n, hint=hint))
        assert torch.equal(got, ref) and torch.equal(got, one)


def test_one_call_on_a_ragged_graph_builds_the_generic_plan():
    """not the reference layout: in-degrees 0 .. dozens, duplicate edges, self loops -> the in-call plan falls through to the generic builder"""
    rng = np.random.default_rng(5)
    n, E = 3000, 14000
    ei = torch.from_numpy(np.stack([rng.integers(0, n, E), rng.integers(0, n // 2, E)]).astype(np.int64)).to(DEV)
    g = torch.Generator().manual_seed(1)
    data = Config(x=torch.randn(n, 29, generator=g).to(DEV), edge_attr=torch.randn(E, 20, generator=g).to(DEV), edge_index=ei)
    net = hip_static()
    one, ref = _both_paths(net, data)
    assert torch.equal(one, ref)
    with torch.no_grad():
        want = oracle_static().inference_layer(Config(x=data.x.cpu(), edge_attr=data.edge_attr.cpu(), edge_index=ei.cpu()))
    assert (one.cpu() - want).abs().max().item() <= TOL_LOGIT * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16x3f", "f16x2d"])
def test_one_call_in_the_other_arithmetic_modes(mode):
    from dgnn_amd import ops
    n, x, ea, ei = _scene(700, seed=8)
    net = hip_static()
    data = Config(x=x.to(DEV), edge_attr=ea.to(DEV), edge_index=ei)
    old = ops.GEMM_MODE
    ops.GEMM_MODE = ops.GEMM_MODE_NAMES[mode]
    try:
        one, ref = _both_paths(net, data)
    finally:
        ops.GEMM_MODE = old
    assert torch.equal(one, ref)


def test_one_call_follows_the_weights_and_other_widths():
    """an optimizer step / load_state_dict between calls must show (folds and prepared parameters are cached against the tensors' versions);
    widths whose last layer cannot carry the decoder run layer + decoder apart inside the call; unsupported widths fall back to the per-layer path"""
    n, x, ea, ei = _scene(500, seed=2)
    data = Config(x=x.to(DEV), edge_attr=ea.to(DEV), edge_index=ei)
    net = hip_static()
    a = net.inference_layer(data)
    with torch.no_grad():
        net.convs[2][0].lin_j.weight.mul_(1.5)
        net.decoder[1].module.running_var.mul_(2.0)
    b, ref = _both_paths(net, data)
    assert torch.equal(b, ref) and not torch.equal(a, b)
    for convs in ((64, 64, 128, 128), (64, 64, 64, 128), (64, 128, 256, 128)):       # the last: 128 -> 256 has no fused kernel -> per-layer path
        sd = oracle_static(convs=convs, load=False, seed=4).state_dict()
        net2 = hip_static(convs=convs, sd=sd)
        one, ref = _both_paths(net2, data)
        assert torch.equal(one, ref)
        with torch.no_grad():
            want = oracle_static(convs=convs, load=False, seed=4).inference_layer(Config(x=x, edge_attr=ea, edge_index=ei.cpu()))
        assert (one.cpu() - want).abs().max().item() <= TOL_LOGIT * max(1.0, want.abs().max().item())


def test_static_infer_c_abi_plain_call():
    """the entry point as a C caller uses it: no prepared buffers, a caller-owned plan with eid == NULL (edge rows already in plan order), BatchNorm folded
    by the caller -- against the oracle"""
    from dgnn_amd import ops
    from dgnn_amd._lib import check, lib, ptr
    n, x, ea, ei = _scene(800, seed=6, layout="contiguous")
    onet = oracle_static()
    with torch.no_grad():
        want = onet.inference_layer(Config(x=x, edge_attr=ea, edge_index=ei.cpu()))
    sd = {k: v.to(DEV) for k, v in onet.state_dict().items()}
    rowptr, src, eid = ops.plan_build(ei, n, by=1)
    ea_sorted = ea.to(DEV)[eid.long()].contiguous()
    xd = x.to(DEV)[:, 1:].contiguous()
    L = 4
    fold = lambda p: ops.bn_fold(sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"], 1e-5)
    folds = [fold("convs.%d.norm.module" % i) for i in range(L)]
    s1, h1 = fold("decoder.1.module")
    arr = lambda ts: (C.c_void_p * L)(*[t.data_ptr() for t in ts])
    widths = (C.c_int32 * (L + 1))(28, 64, 128, 128, 128)
    logits = torch.empty((n, 2), dtype=torch.float32, device=DEV)
    work = torch.empty(int(lib().dgnn_static_infer_workspace_bytes(n, L, widths)), dtype=torch.uint8, device=DEV)
    keep = [[sd["convs.%d.conv.%s" % (i, k)] for i in range(L)] for k in ("lin_e.weight", "lin_e.bias", "lin_j.weight", "lin_j.bias", "lin_i.weight")]
    rc = lib().dgnn_static_infer_fwd(None, 0, 0, 4 * n, 0, ptr(rowptr), ptr(src), None, None, n, ptr(xd), 28, ptr(ea_sorted), 20, 20, L, widths, arr(keep[0]), arr(keep[1]),
                                     arr(keep[2]), arr(keep[3]), arr(keep[4]), arr([f[0] for f in folds]), arr([f[1] for f in folds]), None,
                                     ptr(sd["decoder.0.weight"]), ptr(sd["decoder.0.bias"]), ptr(s1), ptr(h1), 64, ptr(sd["decoder.3.weight"]), ptr(sd["decoder.3.bias"]), 2,
                                     1, ops.GEMM_F16X2, ptr(work), ptr(logits), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    check(rc, "dgnn_static_infer_fwd")
    assert (logits.cpu() - want).abs().max().item() <= TOL_LOGIT
    # a width the fused kernels do not have: refused before anything is launched
    bad = (C.c_int32 * (L + 1))(28, 64, 128, 256, 128)
    logits.fill_(7.0)
    rc = lib().dgnn_static_infer_fwd(None, 0, 0, 4 * n, 0, ptr(rowptr), ptr(src), None, None, n, ptr(xd), 28, ptr(ea_sorted), 20, 20, L, bad, arr(keep[0]), arr(keep[1]),
                                     arr(keep[2]), arr(keep[3]), arr(keep[4]), arr([f[0] for f in folds]), arr([f[1] for f in folds]), None,
                                     ptr(sd["decoder.0.weight"]), ptr(sd["decoder.0.bias"]), ptr(s1), ptr(h1), 64, ptr(sd["decoder.3.weight"]), ptr(sd["decoder.3.bias"]), 2,
                                     1, ops.GEMM_F16X2, ptr(work), ptr(logits), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == ops.DGNN_E_UNSUPPORTED
    torch.cuda.synchronize()
    assert (logits == 7.0).all()


def test_ignatius_full_scene_one_call_vs_reference_logits():
    from test_gpu_scale import logit_check
    g = gold("static_f4_ignatius_full.npz")
    n = g["x"].shape[0]
    fg = np.random.default_rng(int(g["fgeom_seed"])).standard_normal((4 * n, 4)).astype(np.float32)
    ea = torch.from_numpy(np.concatenate([fg, g["edge_attr16"]], axis=1)).to(DEV)
    pairs = np.stack([np.repeat(np.arange(n, dtype=np.int64), 4), g["adj_dst"].astype(np.int64)], 1)
    data = Config(x=torch.from_numpy(g["x"]).to(DEV), edge_attr=ea, edge_index=torch.from_numpy(pairs).to(DEV).t())
    net = hip_static()
    one, ref = _both_paths(net, data)
    assert torch.equal(one, ref)
    logit_check(one.cpu().numpy(), g["logits"], g["logits64"])


def test_ignatius_layers_repeat_bit_for_bit_with_cold_caches():
    """Run-to-run determinism of the wave-specialised 128 -> 128 launches (plain and decoder-carrying) where it once failed: the Ignatius scene has
    67017 cells = 2094 tiles of 32 + 9, so one workgroup owns a last tile in which five of the eight producer groups are empty; those groups' `ready`
    counts used to land before the slot's previous tile was fully produced, and with cold caches (a 1 GiB sweep between launches: gathers of some quads
    take microseconds) the consumers read a late quad's rows too early -- one launch in five, always tile 2030.  (tools/det_ws_real.py locates it.)"""
    from dgnn_amd.graph import GraphPlan
    g = gold("static_f4_ignatius_full.npz")
    n = g["x"].shape[0]
    assert n % 32 == 9
    fg = np.random.default_rng(int(g["fgeom_seed"])).standard_normal((4 * n, 4)).astype(np.float32)
    ea = torch.from_numpy(np.concatenate([fg, g["edge_attr16"]], axis=1)).to(DEV)
    pairs = np.stack([np.repeat(np.arange(n, dtype=np.int64), 4), g["adj_dst"].astype(np.int64)], 1)
    ei = torch.from_numpy(pairs).to(DEV).t().contiguous()
    net = hip_static()
    plan = GraphPlan(ei, n, n)
    acts = [torch.from_numpy(g["x"]).to(DEV)[:, 1:].contiguous()]
    for i in range(3):
        acts.append(net._eval_layers(acts[-1], n, ea, [plan] * 4, True, only=i).clone())
    junk = torch.empty(1 << 28, device=DEV)
    for i, decode in ((2, False), (3, False), (3, True)):
        first = net._eval_layers(acts[i], n, ea, [plan] * 4, True, only=i, decode=decode).clone()
        for rep in range(60):
            junk.add_(1.0)
            o = net._eval_layers(acts[i], n, ea, [plan] * 4, True, only=i, decode=decode)
            bad = (o != first).any(1).nonzero().flatten()
            assert bad.numel() == 0, (i, decode, rep, bad[:8].tolist())


# ---- one rank's part of a partitioned scene in one call, the library's RCCL exchange inside it -----------------------------------------------------
def _self_halo_worker(rank, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    from dgnn_amd.partition import HaloExchange, PartitionedScene, build_self_halo_part
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    net = hip_static()
    for points, frac in ((3000, 0.04), (700, 0.5), (2000, 0.0)):
        adj, cent, _ = delaunay_tet_graph(points, seed=points)
        n = adj.shape[0] // 4
        ei = adj.T.astype(np.int64)
        x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
        ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
        full = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(ei).to(DEV)))
        # cells in a slab of the scene are ALSO held as halo rows, filled by the library's exchange from the rank's own rows (its own peer)
        lo = np.quantile(cent[:, 0], 0.5 - frac / 2)
        hi = np.quantile(cent[:, 0], 0.5 + frac / 2)
        remote = np.nonzero((cent[:, 0] >= lo) & (cent[:, 0] < hi))[0] if frac else np.zeros(0, np.int64)
        lp = build_self_halo_part(ei, n, remote)
        assert lp.n_halo == remote.shape[0] and (frac == 0 or 0 < lp.n_interior < lp.n_own)
        rows = np.concatenate([lp.own_gid, lp.halo_gid])
        scene = PartitionedScene(lp, x[torch.from_numpy(rows).to(DEV)], ea[torch.from_numpy(lp.edge_gid).to(DEV)], DEV)
        # the halo's INPUT rows are resident (layer 0 needs no exchange); every later layer's halo rows arrive through RCCL
        assert type(scene.exchange) is HaloExchange and (scene.exchange._native is not None) == (frac > 0)
        gid = torch.from_numpy(lp.own_gid).to(DEV)
        for one_call in (True, False, True):
            scene.one_call = one_call
            for rebuild in (True, False):
                logits = scene.inference_layer(net, rebuild_plan=rebuild)
                assert scene.used_one_call == one_call
                got = torch.empty_like(full)
                got[gid] = logits
                assert torch.equal(got, full), (points, frac, one_call, rebuild)
    torch.cuda.synchronize()
    open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.destroy_process_group()


def test_partitioned_one_call_with_rccl_self_exchange_equals_whole_scene(tmp_path):
    """dgnn_static_infer_partitioned_fwd on a part that is its own peer: plan of the local graph, interior / boundary launches, the library's RCCL exchange
    between the layers, decoder-carrying last launches -- bit-identical to the whole scene, and to the per-layer chain it replaces."""
    import socket
    import torch.multiprocessing as mp
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    mp.spawn(_self_halo_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))


# ---- ring parts: a scene cut across ranks with NO exchange (every rank recomputes the rings of halo cells later layers read) -------------------------
def test_ring_parts_of_a_tiny_scene_cover_it_and_agree():
    """more ranks than a scene has room for: rings swallow the whole scene, a rank may own a handful of cells or none -- the union still equals the whole"""
    from dgnn_amd.partition import PartitionedScene, build_ring_part, rcb_partition
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, cent, _ = delaunay_tet_graph(9, seed=2)
    n = adj.shape[0] // 4
    ei = adj.T.astype(np.int64)
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    net = hip_static()
    full = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(ei).to(DEV)))
    world = 16
    part = rcb_partition(cent, world)
    part[part == 3] = 4                     # rank 3 owns nothing
    got = torch.full_like(full, float("nan"))
    for rank in range(world):
        lp = build_ring_part(ei, part, rank, world, net.num_layers)
        rows = np.concatenate([lp.own_gid, lp.halo_gid])
        scene = PartitionedScene(lp, x[torch.from_numpy(rows).to(DEV)], ea[torch.from_numpy(lp.edge_gid).to(DEV)], DEV)
        logits = scene.inference_layer(net)
        assert logits.shape == (lp.n_own, 2) and (rank != 3 or lp.n_own == 0)
        got[torch.from_numpy(lp.own_gid).to(DEV)] = logits
    assert torch.equal(got, full)


@pytest.mark.parametrize("world,storage", [(2, "f32"), (8, "f32"), (3, "bf16")])
def test_ring_parts_union_equals_whole_scene(world, storage):
    """dgnn_static_infer_rings_fwd / the per-layer chain over the same destination prefixes: rank after rank on the one GPU (the ranks are independent:
    this IS the multi-GPU computation), union of the logits == the whole scene's, bit for bit."""
    from dgnn_amd import ops
    from dgnn_amd.partition import PartitionedScene, build_ring_part, rcb_partition, ring_dst
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal, loader_cell_order
    adj, cent, _ = delaunay_tet_graph(5000, seed=11)
    adj, cent, _ = loader_cell_order(adj, cent)
    n = adj.shape[0] // 4
    ei = adj.T.astype(np.int64)
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    net = hip_static()
    if storage == "bf16":
        net.set_storage_dtype(torch.bfloat16)
    full = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(ei).to(DEV)))
    part = rcb_partition(cent, world)
    for one_call in (True, False):
        got = torch.full_like(full, float("nan"))
        for rank in range(world):
            lp = build_ring_part(ei, part, rank, world, net.num_layers)
            assert 0 < lp.n_halo and ring_dst(lp, 4)[3] == lp.n_own
            rows = np.concatenate([lp.own_gid, lp.halo_gid])
            scene = PartitionedScene(lp, x[torch.from_numpy(rows).to(DEV)], ea[torch.from_numpy(lp.edge_gid).to(DEV)], DEV)
            scene.one_call = one_call
            for rebuild in (True, False):
                logits = scene.inference_layer(net, rebuild_plan=rebuild)
                # (the bf16 rings call exists in the decoder-carrying form only: DGNN_FUSE_DECODER=0 runs the per-layer chain either way)
                assert scene.used_one_call == (one_call and (storage != "bf16" or ops.FUSE_DECODER)) and logits.shape == (lp.n_own, 2)
            got[torch.from_numpy(lp.own_gid).to(DEV)] = logits
        assert torch.equal(got, full), (world, storage, one_call)


@pytest.mark.parametrize("extra,storage", [(1, "f32"), (2, "f32"), (1, "bf16")])
def test_ring_parts_built_with_more_rings_than_the_model_has_layers(extra, storage):
    """a part built with hops > num_layers (PartitionedScene.build_synthetic's default is 4 under any model) lists in-edges of rings the stack
    never computes; the plan must be built over the prefix of edges whose destinations layer 0 computes (trusted AND verified builder, one call and
    per-layer chain) -- union of the logits == the whole scene's, bit for bit."""
    import dgnn_amd.partition as P
    from dgnn_amd.partition import PartitionedScene, build_ring_part, rcb_partition, ring_dst
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal, loader_cell_order
    adj, cent, _ = delaunay_tet_graph(4000, seed=13)
    adj, cent, _ = loader_cell_order(adj, cent)
    n = adj.shape[0] // 4
    ei = adj.T.astype(np.int64)
    x = hashed_normal(np.arange(n), 29, seed=3, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=4, device=DEV)
    net = hip_static()
    if storage == "bf16":
        net.set_storage_dtype(torch.bfloat16)
    full = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(ei).to(DEV)))
    world = 3
    part = rcb_partition(cent, world)
    old = P.RING_TRUSTED_PLAN
    try:
        for trusted in (True, False):
            P.RING_TRUSTED_PLAN = trusted
            for one_call in (True, False):
                got = torch.full_like(full, float("nan"))
                for rank in range(world):
                    lp = build_ring_part(ei, part, rank, world, net.num_layers + extra)
                    assert len(lp.ring_counts) == net.num_layers + extra and ring_dst(lp, net.num_layers)[-1] == lp.n_own
                    assert lp.edge_index.shape[1] > int(np.searchsorted(lp.edge_index[1], ring_dst(lp, net.num_layers)[0]))   # surplus edges exist
                    rows = np.concatenate([lp.own_gid, lp.halo_gid])
                    scene = PartitionedScene(lp, x[torch.from_numpy(rows).to(DEV)], ea[torch.from_numpy(lp.edge_gid).to(DEV)], DEV)
                    scene.one_call = one_call
                    for rebuild in (True, False):
                        logits = scene.inference_layer(net, rebuild_plan=rebuild)
                    got[torch.from_numpy(lp.own_gid).to(DEV)] = logits
                assert torch.equal(got, full), (extra, storage, trusted, one_call)
    finally:
        P.RING_TRUSTED_PLAN = old


def test_static_infer_rings_c_abi_rejects_growing_destination_counts():
    from dgnn_amd import ops
    from dgnn_amd._lib import lib, ptr
    L = 4
    widths = (C.c_int32 * (L + 1))(28, 64, 128, 128, 128)
    nd = (C.c_int64 * L)(10, 12, 8, 8)
    t = torch.zeros(64, device=DEV)
    arr = (C.c_void_p * L)(*[t.data_ptr()] * L)
    i32 = torch.zeros(64, dtype=torch.int32, device=DEV)
    rc = lib().dgnn_static_infer_rings_fwd(None, 0, 0, 40, 1, ptr(i32), ptr(i32), None, None, 1, 16, nd, ptr(t), 28, ptr(t), 20, 20, L, widths, arr, arr, arr, arr, arr,
                                           arr, arr, None, None, None, None, None, 0, None, None, 0, 0, ops.GEMM_F16X2, ptr(t), ptr(t), None)
    assert rc == -1      # DGNN_E_INVALID (include/dgnn_hip.h)


@pytest.mark.parametrize("points,unsigned", [(7, True), (600, True), (9000, True), (9000, False)])
def test_one_call_bf16_storage_equals_the_per_layer_path(points, unsigned):
    """dgnn_static_infer_rings_fwd_bf16 on a whole scene: the chain of the per-layer bf16 entry points (fp32 input rows read in place, 16-bit rows between
    the layers -- unsigned or plain bf16 --, decoder in the last launch), bit for bit; other bf16 arithmetic: the call declines, same logits."""
    from dgnn_amd import ops
    if not ops.FUSE_DECODER:
        pytest.skip("the bf16 whole-scene call exists in the decoder-carrying form only (DGNN_FUSE_DECODER=0: the per-layer chain)")
    n, x, ea, ei = _scene(points, seed=points + 1)
    net = hip_static()
    net.set_storage_dtype(torch.bfloat16)
    data = Config(x=x.to(DEV), edge_attr=ea.to(DEV), edge_index=ei)
    old = ops.BF16_UNSIGNED_ROWS
    ops.BF16_UNSIGNED_ROWS = unsigned
    try:
        assert net._one_call_tables(data.x[:, 1:], data.edge_attr) is not None
        one, ref = _both_paths(net, data)
        assert one.shape == (n, 2) and one.dtype == torch.float32 and torch.equal(one, ref)
        mode = ops.BF16_MODE
        ops.BF16_MODE = ops.BF16_SINGLE
        try:
            assert net._one_call_tables(data.x[:, 1:], data.edge_attr) is None
            a, b = _both_paths(net, data)
            assert torch.equal(a, b)
        finally:
            ops.BF16_MODE = mode
    finally:
        ops.BF16_UNSIGNED_ROWS = old
    # fp32 storage again on the same model object: the cached tables follow the storage type
    net.set_storage_dtype(torch.float32)
    one32, ref32 = _both_paths(net, data)
    assert torch.equal(one32, ref32) and (one32 - one).abs().max().item() < 0.2


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_inference_layer_on_an_empty_and_a_single_cell_scene(storage):
    """no cells -> logits [0, 2]; one cell whose four neighbour slots point at itself (the smallest reference-layout scene) -> finite logits equal to the
    per-layer path's; one call and per-layer chain alike"""
    net = hip_static()
    if storage == "bf16":
        net.set_storage_dtype(torch.bfloat16)
    empty = Config(x=torch.zeros(0, 29, device=DEV), edge_attr=torch.zeros(0, 20, device=DEV), edge_index=torch.zeros(2, 0, dtype=torch.int64, device=DEV))
    one, ref = _both_paths(net, empty)
    assert one.shape == (0, 2) and ref.shape == (0, 2)
    g = torch.Generator().manual_seed(1)
    single = Config(x=torch.randn(1, 29, generator=g).to(DEV), edge_attr=torch.randn(4, 20, generator=g).to(DEV),
                    edge_index=torch.zeros(2, 4, dtype=torch.int64, device=DEV))
    one, ref = _both_paths(net, single)
    assert one.shape == (1, 2) and torch.isfinite(one).all() and torch.equal(one, ref)


def test_ring_parts_of_a_ragged_graph():
    """a graph that is NOT 4-regular (a tenth of the edges removed, some cells without in-edges): rings follow the in-edges that exist, every layer is still a
    destination prefix of one plan.  On such a graph the fused kernels' generic path sums a cell's messages in an association that depends on where its edges
    fall in the tile's 16-edge blocks, so whole scene and parts agree to fp32 rounding (measured 2.9e-6), not bit for bit as on reference-layout scenes;
    the one call and the per-layer chain of the SAME part are bit-identical."""
    from dgnn_amd.partition import PartitionedScene, build_ring_part, rcb_partition
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, cent, _ = delaunay_tet_graph(2500, seed=21)
    n = adj.shape[0] // 4
    keep = np.random.default_rng(5).random(4 * n) > 0.1
    keep[(adj[:, 1] % 50) == 7] = False               # cells 7, 57, ... lose all their in-edges
    ei = adj[keep].T.astype(np.int64)
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    ea_all = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    ea = ea_all[torch.from_numpy(np.nonzero(keep)[0]).to(DEV)]
    net = hip_static()
    full = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(ei).to(DEV)))
    part = rcb_partition(cent, 3)
    res = {}
    for one_call in (True, False):
        got = torch.full_like(full, float("nan"))
        for rank in range(3):
            lp = build_ring_part(ei, part, rank, 3, net.num_layers)
            rows = np.concatenate([lp.own_gid, lp.halo_gid])
            scene = PartitionedScene(lp, x[torch.from_numpy(rows).to(DEV)], ea[torch.from_numpy(lp.edge_gid).to(DEV)], DEV)
            scene.one_call = one_call
            got[torch.from_numpy(lp.own_gid).to(DEV)] = scene.inference_layer(net)
            assert scene.used_one_call == one_call
        assert not torch.isnan(got).any() and (got - full).abs().max().item() <= 1e-5, one_call
        res[one_call] = got
    assert torch.equal(res[True], res[False])
