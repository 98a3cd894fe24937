"""GPU twins of tests/test_trainer_cpu.py (HIP SurfaceNet instead of the oracle) + the RCCL smoke test + generate()."""
import os
import socket

import numpy as np
import pytest
import torch

from dgnn_amd.config import Config
from helpers import gold, oracle_static
from test_gpu_parity import DEV, hip_static
from test_trainer_cpu import blocks, make_clf, small_scene

pytestmark = pytest.mark.gpu


def _free_port():
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    return port


# ---- RCCL: backend "nccl" at world_size 1 (the box has one GPU) -----------------------------------------------------------
def _rccl_worker(rank, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    from dgnn_amd import ops
    from dgnn_amd.partition import HaloExchange, LocalPart, allreduce_gradients
    # (a) the collective of the data-parallel step on RCCL
    t = torch.arange(1000, dtype=torch.float32, device=DEV)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))
    lin = torch.nn.Linear(8, 8).to(DEV)
    lin(torch.randn(4, 8, device=DEV)).sum().backward()
    g0 = lin.weight.grad.clone()
    allreduce_gradients(lin)          # world 1: returns before the collective
    assert torch.equal(g0, lin.weight.grad)
    # (b) the halo exchange code path on RCCL: grouped ncclSend/ncclRecv on the side stream, receive in place into the tail
    # of the activation buffer, interior work queued between start() and wait().  The one rank is its own peer.
    n_own, n_halo, c = 4096, 512, 128
    send_idx = torch.randperm(n_own, generator=torch.Generator().manual_seed(0))[:n_halo].numpy().astype(np.int64)
    lp = LocalPart(rank=0, world=1, n_total=n_own, own_gid=np.arange(n_own), n_interior=n_own - n_halo, halo_gid=send_idx.copy(),
                   edge_index=np.zeros((2, 0), np.int64), edge_gid=np.zeros(0, np.int64), send_idx=send_idx,
                   send_counts=[n_halo], recv_counts=[n_halo])
    for native in (True, False):
        # native: the exchange as LIBRARY calls (dgnn_halo_plan_create / dgnn_halo_exchange_start / _wait, csrc/halo.hip: pack kernel + one RCCL
        # group on the library's side stream, its own communicator made from a unique id); else the torch.distributed transport of rounds 1-3
        os.environ["DGNN_NATIVE_HALO"] = "1" if native else "0"
        ex = HaloExchange(lp, DEV, pack=ops.gather_rows)
        assert ex.active and ex.stream is not None and not ex.via_host and (ex._native is not None) == native
        for it in range(3):
            h = torch.full((n_own + n_halo, c), float("nan"), device=DEV)
            h[:n_own] = torch.randn(n_own, c, device=DEV)
            ex.start(h)
            busy = ops.relu(h[:n_own])            # "interior" work on the compute stream while the rows travel
            ex.wait()
            assert torch.equal(h[n_own:], h[torch.from_numpy(send_idx).to(DEV)]), it
            assert torch.equal(busy, torch.relu(h[:n_own]))
        # 16-bit rows (bf16 storage, and the unsigned rows' int16 container) and another width through the same plan
        for dt, cc in ((torch.bfloat16, 64), (torch.int16, 128)):
            h = torch.zeros((n_own + n_halo, cc), dtype=dt, device=DEV)
            h[:n_own] = (torch.randn(n_own, cc, device=DEV) * 100).to(dt)
            ex.start(h)
            ex.wait()
            assert torch.equal(h[n_own:], h[torch.from_numpy(send_idx).to(DEV)]), dt
        del ex
    # the C ABI directly, with a row stride wider than the row (round 5: one packed message per peer lands in the plan's staging area and is spread
    # into the strided tail by a copy kernel -- what travels never depends on a side's row stride)
    if True:
        import ctypes as C
        from dgnn_amd._lib import check, lib, ptr
        L = lib()
        assert L.dgnn_rccl_available()
        uid = (C.c_ubyte * 128)()
        check(L.dgnn_comm_unique_id(uid), "uid")
        comm, plan = C.c_void_p(), C.c_void_p()
        check(L.dgnn_comm_create(uid, 0, 1, C.byref(comm)), "comm")
        assert L.dgnn_comm_count(comm) == 1 and L.dgnn_comm_count(None) < 0
        idx32 = torch.from_numpy(send_idx.astype(np.int32)).to(DEV)
        one = (C.c_int64 * 1)(n_halo)
        check(L.dgnn_halo_plan_create(0, 1, n_own, ptr(idx32), one, one, C.byref(plan)), "plan")
        assert L.dgnn_halo_send_rows(plan) == n_halo and L.dgnn_halo_recv_rows(plan) == n_halo
        wide = torch.zeros((n_own + n_halo, 40), device=DEV)
        wide[:n_own] = torch.randn(n_own, 40, device=DEV)
        view = wide[:, :28]                           # rows of 28 floats, stride 40
        keep = wide[n_own:, 28:].clone()
        sbuf = torch.empty(n_halo * 28 * 4, dtype=torch.uint8, device=DEV)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(L.dgnn_halo_exchange_start(plan, comm, ptr(view), 40, 28, 4, ptr(sbuf), st), "start")
        check(L.dgnn_halo_exchange_wait(plan, st), "wait")
        torch.cuda.synchronize()
        assert torch.equal(wide[n_own:, :28], wide[torch.from_numpy(send_idx).to(DEV), :28]) and torch.equal(wide[n_own:, 28:], keep)
        check(L.dgnn_halo_plan_destroy(plan), "plan destroy")
        check(L.dgnn_comm_destroy(comm), "comm destroy")
    torch.cuda.synchronize()
    open(os.path.join(out_dir, "ok"), "w").write("rccl ok")
    dist.destroy_process_group()


def test_rccl_backend_loads_allreduce_and_halo_self_exchange(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))


# ---- data-parallel training of the HIP model, two processes on the one GPU under gloo --------------------------------------
def _dp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dgnn_amd.learning.runModel import Metrics, Trainer
    clf = make_clf()
    clf.temp.device = DEV
    clf.training.metrics = Metrics()
    adj, n, x, ea, y = small_scene()
    net = hip_static(train=True)
    opt = torch.optim.Adam(net.parameters(), lr=0.005)
    tr = Trainer(net)
    for step in range(2):
        n_id, adjs = blocks(adj, n, range(40 * rank + 8 * step, 40 * rank + 8 * step + 24))
        tr.train(Config(all=Config(x=x.to(DEV), y=y.to(DEV), edge_attr=ea.to(DEV)), batch_n_id=n_id.to(DEV),
                        batch_adjs=[(a.to(DEV), e.to(DEV), s) for a, e, s in adjs]), opt, clf)
    torch.save({k: v.cpu() for k, v in net.state_dict().items()}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_training_hip_model_two_processes(tmp_path):
    """Replicas of the HIP SurfaceNet stay bit-equal through two all-reduced steps and match the oracle trained the same way
    (mean of the two shards' gradients) within training tolerance."""
    import torch.multiprocessing as mp
    from dgnn_amd.learning.runModel import Metrics, Trainer
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    sd = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(2)]
    names = [k for k, _ in oracle_static().named_parameters()]
    for k in names:
        assert torch.equal(sd[0][k], sd[1][k]), k
    # oracle, single process, hand-averaged gradients
    clf = make_clf()
    adj, n, x, ea, y = small_scene()
    nets = [oracle_static(train=True) for _ in range(2)]
    opts = [torch.optim.Adam(m.parameters(), lr=0.005) for m in nets]
    tr = [Trainer(m) for m in nets]
    for step in range(2):
        for r in range(2):
            clf.training.metrics = Metrics()
            n_id, adjs = blocks(adj, n, range(40 * r + 8 * step, 40 * r + 8 * step + 24))
            d = Config(all=Config(x=x, y=y, edge_attr=ea), batch_n_id=n_id, batch_adjs=adjs)
            logits = nets[r](d)
            n_sup = adjs[-1][2][1]
            d.batch_x, d.batch_gt = x[n_id[:n_sup]], y[n_id[:n_sup]]
            opts[r].zero_grad()
            tr[r].calcLossAndOA(logits, None, d, clf, clf.training.metrics).backward()
        for p0, p1 in zip(nets[0].parameters(), nets[1].parameters()):
            g = (p0.grad + p1.grad) / 2
            p0.grad.copy_(g)
            p1.grad.copy_(g)
        for o in opts:
            o.step()
    for k, p in nets[0].named_parameters():
        # Adam's first steps are sign-like (|update| = lr): compare the update direction where the gradient is not noise
        got, want = sd[0][k], p.detach()
        assert (got - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item()), k


# ---- train_test loop + resume on the HIP model ---------------------------------------------------------------------------
def test_train_test_loop_and_resume_hip(tmp_path):
    from dgnn_amd.learning.runModel import Trainer, load_epoch
    from dgnn_amd.sampler import NeighborSampler
    clf = make_clf(tmp_path)
    clf.temp.device = DEV
    adj, n, x, ea, y = small_scene(600, seed=5)
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    all_ = Config(x=x.to(DEV), y=y.to(DEV), edge_attr=ea.to(DEV))
    loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=torch.arange(0, 3 * 64), num_nodes=n, batch_size=64)
    val = Config(x=all_.x, y=all_.y, edge_attr=all_.edge_attr, edge_index=ei, infinite=torch.zeros(n))
    data = Config(train=Config(all=all_, batches=loader), validation=Config(all=[val], batches=[[]]))
    net = hip_static(train=True)
    rows = Trainer(net).train_test(data, clf)
    assert len(rows) == 4 and rows[-1]["test_best_loss"] <= rows[0]["test_best_loss"]
    assert sorted(os.listdir(os.path.join(str(tmp_path), "models"))) == ["model_1.ptm", "model_2.ptm", "model_3.ptm", "model_best.ptm"]
    net.eval()
    want = net.inference_layer(val)
    clf.training.load_epoch = "3"
    net2 = hip_static(sd=oracle_static(load=False, seed=9).state_dict())
    assert load_epoch(net2, clf)
    assert torch.equal(net2.eval().inference_layer(val), want)
    # the checkpoint is a plain state_dict the reference's run.py:156 loads as is: same keys as the shipped one
    sd = torch.load(os.path.join(str(tmp_path), "models", "model_best.ptm"), map_location="cpu")
    assert list(sd.keys()) == list(oracle_static().state_dict().keys())


# ---- generate(data, prediction, clf) (reference processing/generate_mesh.py:61) ---------------------------------------------
def test_generate_signature_labels_and_interface(tmp_path):
    from dgnn_amd.processing.generate_mesh import generate
    rng = np.random.default_rng(0)
    n, n_inf, F, V = 5000, 300, 9000, 2500
    infinite = np.zeros(n, np.int32)
    infinite[rng.choice(n, n_inf, replace=False)] = 1
    nf = n - n_inf
    nfacets = rng.integers(-1, nf, size=(F, 2)).astype(np.int32)
    os.makedirs(os.path.join(str(tmp_path), "gt"))
    np.savez(os.path.join(str(tmp_path), "gt", "7_3dt.npz"), vertices=rng.random((V, 3)), tetrahedra=rng.integers(0, V, (nf, 4)).astype(np.int32),
             facets=rng.integers(0, V, (F, 3)).astype(np.int32), nfacets=nfacets)
    prediction = torch.randn(n, 2, generator=torch.Generator().manual_seed(1))
    clf = make_clf()
    clf.temp.device, clf.temp.graph_cut, clf.temp.fix_orientation, clf.temp.metrics = DEV, 0, 0, ["loss"]
    data = Config(path=str(tmp_path), gtfile="gt/7", filename="7", id="", category="", infinite=torch.from_numpy(infinite))
    mesh, eval_dict = generate(data, prediction, clf)                 # prediction on the CPU, as Trainer.inference returns it
    # restatement of reference :75, :93-105
    labels = torch.log_softmax(prediction[torch.from_numpy(infinite) == 0], dim=-1).argmax(1).numpy()
    edges = nfacets.copy()
    edges[edges == -1] = labels.shape[0]
    lab = np.append(labels, 1)
    interfaces = [fi for fi, f in enumerate(edges) if lab[f[0]] != lab[f[1]]]
    want_faces = np.load(os.path.join(str(tmp_path), "gt", "7_3dt.npz"))["facets"][interfaces]
    assert isinstance(eval_dict, dict)
    try:
        import trimesh  # noqa: F401
        assert len(mesh.faces) <= len(want_faces)                    # process=True merges / drops degenerate faces
    except ImportError:
        assert np.array_equal(np.asarray(mesh.faces), want_faces)
        out = mesh.export(os.path.join(str(tmp_path), "m.ply"))
        assert os.path.getsize(out) > 24 * V
    # same result with the prediction already on the device
    mesh2, _ = generate(data, prediction.to(DEV), clf)
    assert np.array_equal(np.asarray(mesh2.faces), np.asarray(mesh.faces))


def _same_grad(k, a, b):
    """Gradients of two launch chains over the same kernels: bit for bit -- except the bias of a Linear that a BatchNorm follows (lin_j, the
    decoder's first Linear) when one chain sums dz's columns inside the merged weight-gradient launch (dgnn_linear_wgrad_x3_cat) and the other
    with dgnn_colsum: both sum in fp64, in different orders, and behind a batch-statistics BatchNorm that sum cancels to rounding noise
    (its exact value is 0), so the two roundings of the noise differ.  Bound: a few fp64 ulps of the sum of magnitudes."""
    if k.endswith("lin_j.bias") or k == "decoder.0.bias":
        return (a - b).abs().max().item() <= 1e-9
    return torch.equal(a, b)


def test_composite_train_layer_is_bit_identical_to_the_separate_calls(monkeypatch):
    """dgnn_sage_layer_train_fwd / _bwd (one library call per layer each way) issue the kernels of the separate entry points in
    the same order: logits, every gradient and every BatchNorm buffer must match bit for bit."""
    from dgnn_amd import ops
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    from test_gpu_parity import gold, f3_data
    g = gold("static_f3_train_blocks.npz")
    d = f3_data(g)
    data = Config(all=Config(x=d.all.x.to(DEV), edge_attr=d.all.edge_attr.to(DEV)), batch_n_id=d.batch_n_id.to(DEV),
                  batch_adjs=[(a.to(DEV), e.to(DEV), s) for a, e, s in d.batch_adjs])
    G = torch.from_numpy(g["G"]).to(DEV)

    def run(composite, data, G, whole=True):
        monkeypatch.setattr(ops, "TRAIN_COMPOSITE", composite)
        monkeypatch.setattr(ops, "TRAIN_WHOLE_MODEL", whole)
        net = hip_static(train=True)
        logits = net(data)
        (logits * G).sum().backward()
        return logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}, {k: b.clone() for k, b in net.named_buffers()}

    for case in range(2):
        if case == 1:   # a 4-hop block of the GPU block builder on a larger scene (sampler-made plans, ragged last tiles)
            adj, _, _ = delaunay_tet_graph(3000, seed=4)
            n = adj.shape[0] // 4
            ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
            x = hashed_normal(np.arange(n), 29, seed=5, device=DEV)
            ea = hashed_normal(np.arange(4 * n), 20, seed=6, device=DEV)
            _, n_id, adjs = NeighborSampler(ei, sizes=[-1] * 4, num_nodes=n, batch_size=257).sample(torch.arange(100, 357, device=DEV))
            data = Config(all=Config(x=x, edge_attr=ea), batch_n_id=n_id, batch_adjs=adjs)
            G = hashed_normal(np.arange(257), 2, seed=7, device=DEV)
        lb, gb, bb = run(False, data, G)                       # separate Functions
        for whole in (True, False):                            # all layers in one call each way | one call per layer each way
            la, ga, ba = run(True, data, G, whole)
            assert torch.equal(la, lb)
            for k in ga:
                assert _same_grad(k, ga[k], gb[k]), (case, whole, k, (ga[k] - gb[k]).abs().max().item())
            for k in ba:
                assert torch.equal(ba[k], bb[k]), (case, whole, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c,C", [(28, 28), (64, 64), (6, 10), (128, 128)])
def test_edge_chain_matches_the_dense_statement(dtype, c, C):
    """out / dphi of the chaining kernels == relu(zeros[E_all, C]; [e_cur] = phi)[e_next, :c] and its autograd, written with
    plain torch ops (reference surfaceNetUpdatedEdgeFilters.py:233-241); e_next is NOT a subset of e_cur here."""
    from dgnn_amd import functional as Fn
    g = torch.Generator().manual_seed(3)
    n_edges, n_cur, n_next = 5000, 1700, 900
    perm = torch.randperm(n_edges, generator=g)
    e_cur = perm[:n_cur].to(DEV)
    e_next = torch.cat([perm[200:900], perm[4000:4200]]).to(DEV)      # 700 produced by this layer, 200 not
    phi = torch.randn(n_cur, C, generator=g).to(DEV).to(dtype).requires_grad_(True)
    G = torch.randn(n_next, c, generator=g).to(DEV).to(dtype)
    pos = torch.full((n_edges,), -1, dtype=torch.int32, device=DEV)
    out = Fn.chain_edges(phi, e_cur, e_next, c, pos, relu=True)
    out.backward(G)
    got_g = phi.grad.clone()
    assert bool((pos == -1).all())
    phi2 = phi.detach().clone().requires_grad_(True)
    dense = torch.zeros(n_edges, C, device=DEV, dtype=dtype)
    dense = dense.index_put((e_cur,), phi2)
    ref = torch.relu(dense)[e_next, :c]
    ref.backward(G)
    assert torch.equal(out, ref)
    assert torch.equal(got_g, phi2.grad)
    # ids outside the scene's edge range are skipped and reported
    from dgnn_amd._lib import DgnnError, lib
    torch.cuda.synchronize()
    lib().dgnn_poll_async_error()
    bad = e_next.clone()
    bad[5] = n_edges + 3
    with pytest.raises(DgnnError):
        Fn.chain_edges(phi.detach(), e_cur, bad, c, pos, relu=True)
        torch.cuda.synchronize()
        from dgnn_amd._lib import check
        check(0, "poll", poll=True)
    assert bool((pos == -1).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_updated_sparse_chaining_is_bit_identical_to_the_dense_edge_tensor(monkeypatch, dtype):
    from dgnn_amd.learning import surfaceNetUpdatedEdgeFilters as U
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, _, _ = delaunay_tet_graph(3000, seed=4)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    x = hashed_normal(np.arange(n), 29, seed=5, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=6, device=DEV)
    _, n_id, adjs = NeighborSampler(ei, sizes=[-1] * 4, num_nodes=n, batch_size=257).sample(torch.arange(100, 357, device=DEV))
    G = hashed_normal(np.arange(257), 2, seed=7, device=DEV)
    clf = Config.wrap(dict(training=dict(model_params=[64, 128, 128, 128], model_name="sage+", loss="kl"),
                           features=dict(normalization_feature=1, keep_normalization_feature=0), temp=dict(device=DEV)))
    torch.manual_seed(1)
    sd = U.SurfaceNet(28, clf).state_dict()
    res = []
    for sparse in (True, False):
        monkeypatch.setattr(U, "CHAIN_SPARSE", sparse)
        net = U.SurfaceNet(28, clf)
        net.load_state_dict(sd)
        net = net.to(DEV).set_storage_dtype(dtype)
        logits = net(Config(x=x, edge_attr=ea, n_id=n_id, adjs=adjs))
        (logits * G).sum().backward()
        res.append((logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize("norm", [None, "log", "sqrt"])
@pytest.mark.parametrize("n", [1, 777, 2048, 300000])
def test_fused_kl_cell_loss_matches_the_reference_op_chain(monkeypatch, norm, n):
    """Trainer.calcLossAndOA through dgnn_kl_cell_loss_{fwd,bwd} vs the reference's own op chain (learning/runModel.py:171-209,
    run here with torch ops on the same GPU tensors): loss, d loss / d logits, and the three running sums."""
    from dgnn_amd.learning import runModel as R
    from test_trainer_cpu import make_clf
    g = torch.Generator().manual_seed(n)
    logits = (3 * torch.randn(n, 2, generator=g)).to(DEV)
    p = torch.rand(n, 1, generator=g)
    p[::7] = 0.0                                   # exact zeros and ones in the target: the xlogy branch
    gt = torch.cat([p, 1 - p, torch.zeros(n, 2)], 1).to(DEV)
    bx = torch.cat([torch.rand(n, 1, generator=g) * 5 + 1e-3, torch.randn(n, 3, generator=g)], 1).to(DEV)
    clf = make_clf()
    clf.regularization.cell_norm = norm
    out = []
    for fused in (True, False):
        monkeypatch.setattr(R, "FUSED_KL_LOSS", fused)
        lg = logits.clone().requires_grad_(True)
        m = R.Metrics()
        loss = R.Trainer(None).calcLossAndOA(lg, None, Config(batch_gt=gt, batch_x=bx), clf, m)
        (loss * 1.7).backward()
        out.append((loss.item(), lg.grad.clone(), m.getCellLoss(), m.getOA(), m.OA_sum, m.samples_sum))
    (la, ga, ca, oa, osa, sa), (lb, gb, cb, ob, osb, sb) = out
    assert abs(la - lb) <= 2e-6 * max(abs(lb), 1e-3), (la, lb)
    assert (ga - gb).abs().max().item() <= 2e-6 * gb.abs().max().item() + 1e-12
    assert abs(ca - cb) <= 2e-6 * abs(cb) and osa == osb and sa == sb == n and oa == ob


@pytest.mark.parametrize("mode", [True, "thread"])
def test_prefetching_block_builder_yields_the_same_blocks_and_plans(mode):
    """iteration with the builder one block ahead (side stream / worker thread, all hops + transposed plans in one library
    call) == the in-line builder, block for block; the transposed plans equal dgnn_plan_build's on the same edge list"""
    from dgnn_amd import ops
    from dgnn_amd.graph import plan_for
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph
    adj, _, _ = delaunay_tet_graph(4000, seed=9)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:1000].to(DEV)
    ref = list(NeighborSampler(ei, sizes=[-1] * 3, node_idx=idx, num_nodes=n, batch_size=128, prefetch=False))
    got = list(NeighborSampler(ei, sizes=[-1] * 3, node_idx=idx, num_nodes=n, batch_size=128, prefetch=mode))
    assert len(ref) == len(got) == 8
    for (ba, na, aa), (bb, nb, ab) in zip(ref, got):
        assert ba == bb and torch.equal(na, nb)
        for (e1, i1, s1), (e2, i2, s2) in zip(aa, ab):
            assert s1 == s2 and torch.equal(e1, e2) and torch.equal(i1, i2)
            plan = plan_for(e2, s2[0], s2[1])
            assert plan._t is not None and plan.edge_rows is not None      # came with the block
            t_ref = ops.plan_build(e2, s2[0], by=0, hint=ops.PLAN_HINT_GENERIC, n_other=s2[1])
            for a, b in zip(plan.transposed, t_ref):
                assert torch.equal(a, b)
            assert torch.equal(plan.edge_rows.long(), i2) and torch.equal(plan.transposed_edge_rows.long(), i2[t_ref[2].long()])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["sage+", "sage"])
def test_updated_composite_layer_matches_the_separate_calls(monkeypatch, dtype, name):
    """dgnn_sage_updated_train_fwd / _bwd (one library call per conv each way) vs the chain of separate Functions: fp32 is
    bit-identical (same kernels, same order); bf16 storage differs only where the composite adds dz.Wr into dx before rounding
    to bf16 instead of after (one rounding instead of two), so it is compared at bf16 resolution."""
    from dgnn_amd import ops
    from dgnn_amd.learning import surfaceNetUpdatedEdgeFilters as U
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, _, _ = delaunay_tet_graph(3000, seed=4)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    x = hashed_normal(np.arange(n), 29, seed=5, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=6, device=DEV)
    _, n_id, adjs = NeighborSampler(ei, sizes=[-1] * 4, num_nodes=n, batch_size=257).sample(torch.arange(100, 357, device=DEV))
    G = hashed_normal(np.arange(257), 2 if name == "sage+" else 128, seed=7, device=DEV)
    clf = Config.wrap(dict(training=dict(model_params=[64, 128, 128, 128], model_name=name, loss="kl"),
                           features=dict(normalization_feature=1, keep_normalization_feature=0), temp=dict(device=DEV)))
    torch.manual_seed(1)
    sd = U.SurfaceNet(28, clf).state_dict()
    res = []
    for composite in (True, False):
        monkeypatch.setattr(ops, "TRAIN_COMPOSITE", composite)
        net = U.SurfaceNet(28, clf)
        net.load_state_dict(sd)
        net = net.to(DEV).set_storage_dtype(dtype)
        logits = net(Config(x=x, edge_attr=ea, n_id=n_id, adjs=adjs))
        (logits * G).sum().backward()
        res.append((logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
    (la, ga), (lb, gb) = res
    assert torch.equal(la, lb)                      # forward: identical in both storage types
    for k in ga:
        if dtype == torch.float32:
            assert torch.equal(ga[k], gb[k]), k
        else:
            assert (ga[k] - gb[k]).abs().max().item() <= 2e-2 * gb[k].abs().max().item() + 1e-6, (k, (ga[k] - gb[k]).abs().max().item(), gb[k].abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["sage+", "sage"])
def test_updated_conv_stack_call_is_bit_identical_to_the_per_layer_calls(monkeypatch, dtype, name):
    """dgnn_updated_stack_fwd / _bwd (all conv layers and the edge chaining per call) issue the per-layer composite calls and the chaining kernels
    in the per-layer path's order: logits and every gradient bit for bit, fp32 and bf16 storage; the chaining table is left all -1."""
    from dgnn_amd import ops
    from dgnn_amd.learning import surfaceNetUpdatedEdgeFilters as U
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, _, _ = delaunay_tet_graph(3000, seed=4)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    x = hashed_normal(np.arange(n), 29, seed=5, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=6, device=DEV)
    _, n_id, adjs = NeighborSampler(ei, sizes=[-1] * 4, num_nodes=n, batch_size=257).sample(torch.arange(100, 357, device=DEV))
    G = hashed_normal(np.arange(257), 2 if name == "sage+" else 128, seed=7, device=DEV)
    clf = Config.wrap(dict(training=dict(model_params=[64, 128, 128, 128], model_name=name, loss="kl"),
                           features=dict(normalization_feature=1, keep_normalization_feature=0), temp=dict(device=DEV)))
    torch.manual_seed(1)
    sd = U.SurfaceNet(28, clf).state_dict()
    res = []
    for stack in (True, False):
        monkeypatch.setattr(ops, "UPDATED_STACK", stack)
        net = U.SurfaceNet(28, clf)
        net.load_state_dict(sd)
        net = net.to(DEV).set_storage_dtype(dtype)
        used = []
        orig = ops.updated_stack_fwd
        monkeypatch.setattr(ops, "updated_stack_fwd", lambda *a, **k: (used.append(1), orig(*a, **k))[1])
        logits = net(Config(x=x, edge_attr=ea, n_id=n_id, adjs=adjs))
        (logits * G).sum().backward()
        monkeypatch.setattr(ops, "updated_stack_fwd", orig)
        assert bool(used) == (stack and ops.TRAIN_COMPOSITE and U.CHAIN_SPARSE)
        assert int((net._chain_pos != -1).sum()) == 0 if net._chain_pos is not None else True
        res.append((logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
    (la, ga), (lb, gb) = res
    assert torch.equal(la, lb)
    for k in ga:
        if k.startswith("out_net") and k.endswith("bias"):
            # the output network's bias sums ride in the merged weight-gradient launch inside the call (fp64, another order than dgnn_colsum's)
            assert (ga[k] - gb[k]).abs().max().item() <= 1e-6 * gb[k].abs().max().item(), k
        else:
            assert torch.equal(ga[k], gb[k]), (k, (ga[k] - gb[k]).abs().max().item())


def test_eval_after_training_forwards_sees_the_new_running_statistics():
    """The library updates BatchNorm's running buffers through raw addresses; their version counters are bumped so that the eval-mode fold (cached
    per BatchNorm against the versions) is rebuilt: eval after train-mode forwards WITHOUT an optimizer step must follow the new statistics."""
    from test_gpu_parity import gold, f3_data
    g = gold("static_f3_train_blocks.npz")
    d = f3_data(g)
    data = Config(all=Config(x=d.all.x.to(DEV), edge_attr=d.all.edge_attr.to(DEV)), batch_n_id=d.batch_n_id.to(DEV),
                  batch_adjs=[(a.to(DEV), e.to(DEV), s) for a, e, s in d.batch_adjs])
    net = hip_static(train=False)
    adj, n, x, ea, y = small_scene(300, seed=3)
    val = Config(x=x.to(DEV), edge_attr=ea.to(DEV), edge_index=torch.from_numpy(adj.T.astype(np.int64)).to(DEV))
    before = net.inference_layer(val).clone()          # fills the fold / prepared caches
    net.train()
    with torch.no_grad():
        for _ in range(3):
            net(data)                                  # running statistics move, no parameter does
    net.eval()
    after = net.inference_layer(val)
    assert not torch.equal(before, after)
    ref = hip_static(train=False)
    ref.load_state_dict(net.state_dict())              # a fresh model with the same tensors has no caches
    assert torch.equal(ref.eval().inference_layer(val), after)


def test_library_adam_follows_torch_adam():
    """dgnn_amd.optim.Adam (one launch per step) against torch.optim.Adam over several steps: same rule (no amsgrad / weight decay), a changing learning
    rate, a parameter that gets no gradient on some steps (its own step count, as in torch), state_dict round trip"""
    from dgnn_amd.optim import Adam
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 28), (64,), (128, 64), (2,), (5000, 3), (1,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*sh, generator=torch.Generator().manual_seed(i)).to(DEV)) for i, sh in enumerate(shapes)]
    pa, pb = mk(), mk()
    oa, ob = Adam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    for step in range(7):
        if step == 4:
            for o in (oa, ob):
                for grp in o.param_groups:
                    grp["lr"] = 3e-4
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 2 and step in (1, 2):          # no gradient for this one on two steps
                a.grad = b.grad = None
                continue
            gr = (torch.randn(*shapes[i], generator=g) * (10.0 ** (i - 3))).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item()) + 1e-7
    assert oa.state[pa[2]]["step"] == 5 and oa.state[pa[0]]["step"] == 7
    assert pa[0]._version >= 7 and pa[2]._version >= 5        # every step bumps the version counters (BatchNorm folds / prepared parameters key on them)
    sd = oa.state_dict()
    oc = Adam(mk(), lr=1.0)
    oc.load_state_dict(sd)
    assert oc.param_groups[0]["lr"] == 3e-4 and oc.state[oc.param_groups[0]["params"][0]]["step"] == 7
    with pytest.raises(TypeError):
        Adam([torch.nn.Parameter(torch.zeros(3))])          # a CPU parameter: the caller keeps torch.optim.Adam


def test_library_adam_takes_over_a_torch_adam_state_mid_run():
    """load_state_dict AFTER steps must switch the launch to the loaded moments (the pointer tables are cached), and a dict saved by
    torch.optim.Adam (`step` a tensor) must load: three steps with torch's Adam, state handed over, four more steps on each -- same parameters"""
    from dgnn_amd.optim import Adam
    shapes = [(64, 28), (64,), (300, 7)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*sh, generator=torch.Generator().manual_seed(i)).to(DEV)) for i, sh in enumerate(shapes)]
    pa, pb = mk(), mk()
    oa, ob = Adam(pa, lr=2e-3), torch.optim.Adam(pb, lr=2e-3)
    g = torch.Generator().manual_seed(5)

    def grads():
        for a, b in zip(pa, pb):
            gr = torch.randn(*a.shape, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
    grads()
    oa.step()                                   # oa has stepped once on its OWN moments (tables cached) ...
    for _ in range(3):
        grads()
        ob.step()
    with torch.no_grad():
        for a, b in zip(pa, pb):
            a.copy_(b)
    import copy
    oa.load_state_dict(copy.deepcopy(ob.state_dict()))     # ... and now takes over torch's state as a checkpoint holds it: `step` tensors, torch's moment layout
    # (a deep copy, as torch.save / torch.load make one: Optimizer.load_state_dict itself keeps the tensors it is handed)
    assert all(isinstance(oa.state[p]["step"], int) and oa.state[p]["step"] == 3 for p in pa)
    for _ in range(4):
        grads()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item()) + 1e-7
    assert oa.state[pa[0]]["step"] == 7
    extra = torch.nn.Parameter(torch.ones(5, device=DEV))
    oa.add_param_group(dict(params=[extra]))    # a new group invalidates the tables too
    extra.grad = torch.ones(5, device=DEV)
    oa.step()
    assert oa.state[extra]["step"] == 1 and (extra < 1).all()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the reference addresses cuda:<n> without set_device)")
def test_library_adam_and_block_builder_follow_their_tensors_device():
    """parameters on cuda:1 while device 0 is current -- the step must launch on device 1's stream"""
    from dgnn_amd.optim import Adam
    assert torch.cuda.current_device() == 0
    d1 = torch.device("cuda:1")
    p = torch.nn.Parameter(torch.ones(1000, device=d1))
    q = torch.nn.Parameter(torch.ones(1000, device=d1))
    oa, ob = Adam([p], lr=1e-2), torch.optim.Adam([q], lr=1e-2)
    for _ in range(3):
        p.grad = torch.full((1000,), 0.5, device=d1)
        q.grad = torch.full((1000,), 0.5, device=d1)
        oa.step()
        ob.step()
    torch.cuda.synchronize(d1)
    assert torch.cuda.current_device() == 0
    assert (p - q).abs().max().item() <= 1e-6
    with pytest.raises(ValueError):
        Adam([torch.nn.Parameter(torch.ones(3, device="cuda:0")), torch.nn.Parameter(torch.ones(3, device=d1))])


def test_direct_training_step_equals_the_autograd_step():
    """Trainer.train without the autograd engine (round 4: SurfaceNet.train_step_direct issues the whole-model calls, the fused loss and its gradient
    directly) against the autograd path (DGNN_TRAIN_DIRECT=0) on the same blocks: loss, every parameter after three Adam steps, BatchNorm buffers,
    metrics -- bit for bit (same kernels in the same order); the gradients of the last step too."""
    import dgnn_amd.learning.runModel as RM
    from dgnn_amd.learning.runModel import Metrics, Trainer
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, _, _ = delaunay_tet_graph(4000, seed=13)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    x[:, 0] = x[:, 0].abs() + 0.05
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
    all_ = Config(x=x, y=torch.cat([occ, 1 - occ], 1), edge_attr=ea)
    out = {}
    for direct in (True, False):
        RM.TRAIN_DIRECT = direct
        try:
            clf = make_clf()
            clf.temp.device = DEV
            clf.temp.current_epoch = 0
            clf.training.metrics = Metrics()
            net = hip_static(train=True)
            tr = Trainer(net)
            opt = RM.make_adam(net.parameters(), 0.005)
            loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=torch.arange(0, 3 * 256, device=DEV), num_nodes=n, batch_size=256)
            losses = []
            for bs, n_id, adjs in loader:
                d = Config(all=all_, batch_n_id=n_id, batch_adjs=adjs)
                losses.append(tr.train(d, opt, clf).item())
                assert d.batch_x.shape == (256, 29) and d.batch_gt.shape == (256, 2)       # left on the data object as the reference does (:273-274)
            out[direct] = (losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, {k: p.grad.clone() for k, p in net.named_parameters()},
                           (clf.training.metrics.getCellLoss(), clf.training.metrics.getOA(), clf.training.metrics.samples_sum))
        finally:
            RM.TRAIN_DIRECT = True
    assert out[True][0] == out[False][0] and out[True][3] == out[False][3]
    for k in out[False][1]:
        assert torch.equal(out[True][1][k], out[False][1][k]), k
    for k in out[False][2]:
        assert torch.equal(out[True][2][k], out[False][2][k]), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_updated_direct_training_step_equals_the_autograd_step(dtype):
    """Round 6: the Updated model's step without the autograd engine (SurfaceNet.train_step_direct: the conv stack's and the output network's library
    calls issued directly around the one-launch kl loss) against forward() + the fused loss + backward() on the same blocks (reference
    learning/surfaceNetUpdatedEdgeFilters.py:216-251, learning/runModel.py:264-282): loss, every gradient and every parameter after three Adam steps,
    bit for bit (same kernels in the same order), in fp32 and in bf16 storage; and the one-launch loss against its two-launch form."""
    import dgnn_amd.learning.runModel as RM
    from dgnn_amd import functional as Fn
    from dgnn_amd import ops
    from dgnn_amd.learning.surfaceNetUpdatedEdgeFilters import SurfaceNet as Updated
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, _, _ = delaunay_tet_graph(4000, seed=13)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    x[:, 0] = x[:, 0].abs() + 0.05
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
    y = torch.cat([occ, 1 - occ], 1)
    ucfg = Config.wrap(dict(training=dict(model_params=[64, 128, 128, 128], model_name="sage+", loss="kl"),
                            features=dict(normalization_feature=1, keep_normalization_feature=0), temp=dict(device=DEV)))
    out = {}
    for direct in (True, False):
        torch.manual_seed(3)
        net = Updated(28, ucfg).to(DEV).train().set_storage_dtype(dtype)
        opt = RM.make_adam(net.parameters(), 0.005)
        loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=torch.arange(0, 3 * 256, device=DEV), num_nodes=n, batch_size=256)
        losses = []
        for bs, n_id, adjs in loader:
            ids = n_id[:bs]
            by, vol = y[ids], x[ids, 0]
            d = Config(x=x, edge_attr=ea, n_id=n_id, adjs=adjs)
            if direct:
                def loss_fn(logits):
                    got = ops.kl_cell_loss_step(logits, by, vol, 0)
                    l2, s2 = ops.kl_cell_loss_fwd(logits, by, vol, 0)               # the one-launch loss == the two-launch loss, bit for bit
                    g2 = ops.kl_cell_loss_bwd(logits, by, vol, 0, s2, torch.ones((), device=DEV))
                    assert torch.equal(got[0], l2) and torch.equal(got[1], s2) and torch.equal(got[2], g2)
                    return got[0], got[2]
                loss = net.train_step_direct(d, loss_fn)
                if loss is None:
                    pytest.skip("this configuration does not take the one-call form (DGNN_TRAIN_COMPOSITE / DGNN_CHAIN_DENSE / DGNN_UPDATED_STACK / "
                                "DGNN_UPDATED_TAIL_IN_CALL select the per-layer calls): train_step_direct steps aside, the Trainer runs forward() + backward()")
            else:
                opt.zero_grad()
                loss, _ = Fn.kl_cell_loss(net(d).float(), by, vol)
                loss.backward()
            opt.step()
            losses.append(loss.item())
        out[direct] = (losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, {k: p.grad.clone() for k, p in net.named_parameters()})
    assert out[True][0] == out[False][0]
    for k in out[False][1]:
        assert torch.equal(out[True][1][k], out[False][1][k]), k
    for k in out[False][2]:
        assert torch.equal(out[True][2][k], out[False][2][k]), k


@pytest.mark.parametrize("model", ["static", "updated"])
def test_block_builder_gathers_the_step_rows_and_the_run_is_the_same(model):
    """NeighborSampler.attach_rows (Trainer.attach_block_rows): x[n_id, 1:], x[ids], y[ids] come out of the block builder -- equal to indexing, for every
    block of a buffer-ring loader and after a second attach; a training run with and without them ends bit-identical (same rows, same kernels)."""
    import dgnn_amd.learning.runModel as RM
    import dgnn_amd.sampler as SM
    from dgnn_amd.learning.runModel import Metrics, Trainer
    from dgnn_amd.sampler import NeighborSampler, block_rows
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    if not SM.ONE_CALL:
        pytest.skip("the block builder gathers the step's rows in its one-call form (DGNN_KHOP_ONE_CALL=0 selects the hop-by-hop builder)")
    adj, _, _ = delaunay_tet_graph(4000, seed=17)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    x[:, 0] = x[:, 0].abs() + 0.05
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
    all_ = Config(x=x, y=torch.cat([occ, 1 - occ], 1), edge_attr=ea)
    out = {}
    for attach in (True, False):
        clf = make_clf()
        clf.temp.device = DEV
        clf.temp.current_epoch = 0
        clf.training.metrics = Metrics()
        if model == "updated":
            from dgnn_amd.learning.surfaceNetUpdatedEdgeFilters import SurfaceNet as Updated
            ucfg = Config.wrap(dict(training=dict(model_params=[64, 128, 128, 128], model_name="sage+", loss="kl"),
                                    features=dict(normalization_feature=1, keep_normalization_feature=0), temp=dict(device=DEV)))
            torch.manual_seed(3)
            net = Updated(28, ucfg).to(DEV).train()
        else:
            net = hip_static(train=True)
        tr = Trainer(net)
        opt = RM.make_adam(net.parameters(), 0.005)
        loader = NeighborSampler(ei, sizes=[-1] * 4, node_idx=torch.arange(0, 5 * 200 + 37, device=DEV), num_nodes=n, batch_size=200, reuse_buffers=True)
        if attach:
            tr.attach_block_rows(loader, all_, net)
            tr.attach_block_rows(loader, all_, net)      # (a second time: buffer sets are rebuilt)
        losses = []
        for bs, n_id, adjs in loader:
            for (src, c0, nc, which) in ((x, 1, 28, "all"), (x, 0, 29, "batch"), (all_.y, 0, 2, "batch")):
                got = block_rows(n_id, src, c0, nc, which)
                assert (got is not None) == attach
                if attach:
                    ids = n_id if which == "all" else n_id[:bs]
                    assert torch.equal(got, src[ids, c0:c0 + nc]), (which, c0)
            d = Config(all=all_, batch_n_id=n_id, batch_adjs=adjs)
            if model == "updated":
                logits = net(Config(x=x, edge_attr=ea, n_id=n_id, adjs=adjs))
                loss = (logits * torch.arange(logits.numel(), device=DEV).view_as(logits).float().cos()).sum()
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(loss.item())
            else:
                losses.append(tr.train(d, opt, clf).item())
                assert d.batch_x.shape == (bs, 29) and d.batch_gt.shape == (bs, 2)
        assert len(losses) == 6
        out[attach] = (losses, {k: v.detach().clone() for k, v in net.state_dict().items()})
    assert out[True][0] == out[False][0]
    for k in out[False][1]:
        assert torch.equal(out[True][1][k], out[False][1][k]), k


def test_aux_stream_backward_gives_identical_gradients():
    """dgnn_train_set_aux_stream(1): weight gradients on the library's second stream beside the dx chain -- same numbers"""
    from dgnn_amd._lib import lib
    from test_gpu_parity import gold, f3_data
    g = gold("static_f3_train_blocks.npz")
    d = f3_data(g)
    data = Config(all=Config(x=d.all.x.to(DEV), edge_attr=d.all.edge_attr.to(DEV)), batch_n_id=d.batch_n_id.to(DEV),
                  batch_adjs=[(a.to(DEV), e.to(DEV), s) for a, e, s in d.batch_adjs])
    G = torch.from_numpy(g["G"]).to(DEV)
    res = []
    was = lib().dgnn_train_set_aux_stream(0)
    try:
        for on in (0, 1, 1):
            lib().dgnn_train_set_aux_stream(on)
            net = hip_static(train=True)
            (net(data) * G).sum().backward()
            torch.cuda.synchronize()
            res.append({k: p.grad.clone() for k, p in net.named_parameters()})
    finally:
        lib().dgnn_train_set_aux_stream(was)
    for k in res[0]:
        assert _same_grad(k, res[0][k], res[1][k]) and torch.equal(res[1][k], res[2][k]), k


def test_block_builder_buffer_ring_gives_the_same_training_run():
    """reuse_buffers=True (blocks built into a ring of three preallocated buffer sets, valid until two more have been drawn) under a
    training loop that consumes each block before asking for the next: same losses and parameters as with fresh tensors per block"""
    from dgnn_amd.learning.runModel import Metrics, Trainer
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    from test_trainer_cpu import make_clf
    adj, _, _ = delaunay_tet_graph(4000, seed=3)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    x[:, 0] = x[:, 0].abs() + 0.05
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
    all_ = Config(x=x, y=torch.cat([occ, 1 - occ], 1), edge_attr=ea)
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:12 * 300 + 77].to(DEV)     # 13 batches, the last one short
    out = []
    for ring in (False, True):
        clf = make_clf()
        clf.temp.device = DEV
        clf.training.metrics = Metrics()
        net = hip_static(train=True)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        tr = Trainer(net)
        losses = []
        for bs, n_id, adjs in NeighborSampler(ei, sizes=[-1] * 4, node_idx=idx, num_nodes=n, batch_size=300, reuse_buffers=ring):
            losses.append(tr.train(Config(all=all_, batch_n_id=n_id, batch_adjs=adjs), opt, clf))
        torch.cuda.synchronize()
        out.append((torch.stack(losses).cpu(), {k: v.detach().clone() for k, v in net.state_dict().items()}))
    assert torch.equal(out[0][0], out[1][0])
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k


def test_static_composite_layer_in_bf16_storage_matches_the_separate_calls(monkeypatch):
    """bf16 storage through dgnn_sage_layer_train_fwd_bf16 / _bwd_bf16 vs the separate bf16 Functions: identical forward; gradients at
    bf16 resolution (the composite adds dz.Wi into dx before the rounding to bf16, the separate path after)"""
    from dgnn_amd import ops
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, _, _ = delaunay_tet_graph(3000, seed=4)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    x = hashed_normal(np.arange(n), 29, seed=5, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=6, device=DEV)
    _, n_id, adjs = NeighborSampler(ei, sizes=[-1] * 4, num_nodes=n, batch_size=257).sample(torch.arange(100, 357, device=DEV))
    data = Config(all=Config(x=x, edge_attr=ea), batch_n_id=n_id, batch_adjs=adjs)
    G = hashed_normal(np.arange(257), 2, seed=7, device=DEV)
    res = []
    for composite in (True, False):
        monkeypatch.setattr(ops, "TRAIN_COMPOSITE", composite)
        net = hip_static(train=True).set_storage_dtype(torch.bfloat16)
        logits = net(data)
        (logits * G).sum().backward()
        res.append((logits.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}, {k: b.clone() for k, b in net.named_buffers()}))
    (la, ga, ba), (lb, gb, bb) = res
    assert torch.equal(la, lb)
    for k in ba:
        assert torch.equal(ba[k], bb[k]), k
    # biases feeding a train-mode BatchNorm have an analytically zero gradient: both paths hold bf16 rounding noise there, so the error is
    # measured against the tensor's own scale plus a floor tied to the largest gradient in the model (as in the fp32 golden test)
    gmax = max(v.abs().max().item() for v in gb.values())
    for k in ga:
        err = (ga[k] - gb[k]).abs().max().item()
        assert err <= 3e-2 * gb[k].abs().max().item() + 2e-3 * gmax, (k, err, gb[k].abs().max().item(), gmax)


def test_buffer_ring_blocks_are_correct_when_consumed_in_order_across_epochs_and_early_exit():
    """reuse_buffers=True: every block, checked at the moment it is handed over, equals the in-line builder's; a loop abandoned mid-epoch
    (the builder thread's job is joined), a second epoch over the same sampler (ring slots reused from the start) and shuffling work too"""
    from dgnn_amd.graph import plan_for
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph
    adj, _, _ = delaunay_tet_graph(5000, seed=21)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(2))[:1500].to(DEV)
    ref = [(b, nid.clone(), [(e.clone(), i.clone(), s) for e, i, s in adjs])
           for b, nid, adjs in NeighborSampler(ei, sizes=[-1] * 3, node_idx=idx, num_nodes=n, batch_size=100, prefetch=False)]
    ring = NeighborSampler(ei, sizes=[-1] * 3, node_idx=idx, num_nodes=n, batch_size=100, reuse_buffers=True)

    def check(k, blk):
        b, nid, adjs = blk
        rb, rn, ra = ref[k]
        assert b == rb and torch.equal(nid, rn)
        for (e1, i1, s1), (e2, i2, s2) in zip(ra, adjs):
            assert s1 == s2 and torch.equal(e1, e2) and torch.equal(i1, i2)
            plan = plan_for(e2, s2[0], s2[1])
            assert torch.equal(plan.edge_rows.long(), i2) and torch.equal(plan.src.long(), e2[0])
            t = plan.transposed
            assert int(t[0][-1]) == e2.size(1)
        # some work on the caller's stream that reads the block while the builder is already filling the next slot
        return float(nid.double().sum().item())

    for k, blk in enumerate(ring):          # epoch 1, abandoned after 7 of 15 blocks
        check(k, blk)
        if k == 6:
            break
    for epoch in range(2):                  # two full epochs over the same sampler
        seen = 0
        for k, blk in enumerate(ring):
            check(k, blk)
            seen += 1
        assert seen == len(ref) == 15


# ---- rows 8f-2 against values the REFERENCE's own learning/runModel.py produced (tests/golden/trainer_f2.npz) -----------------------
@pytest.mark.parametrize("loss,norm", [(l, n) for l in ("kl", "bce", "mse") for n in (None, "log", "sqrt")])
def test_trainer_loss_on_gpu_matches_the_reference_run(loss, norm):
    """Trainer.calcLossAndOA on GPU tensors (kl: the fused dgnn_kl_cell_loss kernels; bce / mse: torch ops on the device) against loss, d loss / d logits
    and the Metrics sums recorded from the reference's Trainer (make_golden.py round3)"""
    from test_reference_host_cpu import check_loss_case
    check_loss_case(gold("trainer_f2.npz"), loss, norm, torch.device(DEV))


def test_trainer_regularizer_on_gpu_matches_the_reference_run():
    from test_reference_host_cpu import check_regularizer
    check_regularizer(gold("trainer_f2.npz"), torch.device(DEV))


def test_trainer_train_steps_hip_model_match_the_reference_run():
    """three Trainer.train steps (block forward / loss / backward / Adam) of the HIP model against the reference's own three steps on the same
    blocks and targets: running cell loss per step (the 2nd and 3rd see the updated weights), OA, and tensors of the final state_dict"""
    from test_reference_host_cpu import check_train_steps
    check_train_steps(gold("trainer_f2.npz"), hip_static(train=True), torch.device(DEV), 2e-4)
