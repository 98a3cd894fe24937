"""Multi-rank path on CPU (gloo, world_size 2 and 3): partition index structures and the per-round halo
exchange, driven with the CPU oracle as the per-layer compute.  The union of the ranks' logits must
equal the single-process oracle result bit for bit (each destination keeps its global edge order)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dgnn_amd.config import Config
from dgnn_amd.partition import HaloExchange, build_local_part, rcb_partition, run_partitioned_layers
from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
from helpers import oracle_static


def _scene(points=500, seed=4):
    adj, cent, _ = delaunay_tet_graph(points, seed)
    n = adj.shape[0] // 4
    x = hashed_normal(np.arange(n), 29, seed=1)
    ea = hashed_normal(np.arange(4 * n), 20, seed=2)
    return adj, cent, x, ea


def test_rcb_and_local_part_structure():
    adj, cent, _, _ = _scene()
    n = adj.shape[0] // 4
    ei = adj.T.astype(np.int64)
    for world in (2, 3, 8):
        part = rcb_partition(cent, world)
        sizes = np.bincount(part, minlength=world)
        assert sizes.min() >= n // world - 1 and sizes.max() <= n // world + world
        lps = [build_local_part(ei, part, r, world) for r in range(world)]
        assert sum(lp.n_own for lp in lps) == n
        for lp in lps:
            # every owned tet keeps exactly its 4 in-edges; the local list is grouped by destination with the global
            # edge order inside a destination (so per-destination summation order == single-process order)
            assert lp.edge_index.shape[1] == 4 * lp.n_own and np.array_equal(lp.edge_index[1], np.repeat(np.arange(lp.n_own), 4))
            assert np.all(np.diff(lp.edge_gid.reshape(-1, 4), axis=1) > 0)
            # interior cells first: they read no halo row and are sent to nobody; boundary cells are the rest
            reads_halo = (lp.edge_index[0] >= lp.n_own).reshape(-1, 4).any(1)
            is_sent = np.zeros(lp.n_own, bool); is_sent[lp.send_idx] = True
            assert 0 < lp.n_interior < lp.n_own
            assert not (reads_halo | is_sent)[:lp.n_interior].any() and (reads_halo | is_sent)[lp.n_interior:].all()
            assert np.all(np.diff(lp.own_gid[:lp.n_interior]) > 0) and np.all(np.diff(lp.own_gid[lp.n_interior:]) > 0)
            assert lp.edge_index[1].max() < lp.n_own and lp.edge_index[0].max() < lp.n_own + lp.n_halo
            gl = np.concatenate([lp.own_gid, lp.halo_gid])
            assert np.array_equal(gl[lp.edge_index[0]], ei[0][lp.edge_gid]) and np.array_equal(gl[lp.edge_index[1]], ei[1][lp.edge_gid])
            assert np.all(part[lp.halo_gid] != lp.rank) and sum(lp.recv_counts) == lp.n_halo
        # what p sends to q is exactly what q expects from p, in the same order
        for p in range(world):
            so = 0
            for q in range(world):
                ns = lps[p].send_counts[q]
                ro = sum(lps[q].recv_counts[:p])
                assert ns == lps[q].recv_counts[p]
                assert np.array_equal(lps[p].own_gid[lps[p].send_idx[so:so + ns]], lps[q].halo_gid[ro:ro + ns])
                so += ns


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ring_part_structure_and_oracle_forward_without_exchange(world):
    """build_ring_part: L rings of halo cells, layer l over the owned cells + rings 1 .. L-1-l, no exchange at all.  Structure, and the CPU oracle's
    layers run over those destination prefixes: the union of the ranks' logits equals the whole scene's."""
    from dgnn_amd.partition import build_ring_part, ring_dst
    adj, cent, x, ea = _scene()
    n = adj.shape[0] // 4
    ei = adj.T.astype(np.int64)
    part = rcb_partition(cent, world)
    torch.set_num_threads(1)
    net = oracle_static()
    L = net.num_layers
    with torch.no_grad():
        ref = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(ei))).numpy()
    got = np.full_like(ref, np.nan)
    for rank in range(world):
        lp = build_ring_part(ei, part, rank, world, L)
        nd = ring_dst(lp, L)
        assert lp.n_interior == lp.n_own and sum(lp.send_counts) == 0 and sum(lp.recv_counts) == 0 and len(lp.ring_counts) == L
        assert np.array_equal(lp.own_gid, np.nonzero(part == rank)[0]) and sum(lp.ring_counts) == lp.n_halo
        assert nd[L - 1] == lp.n_own and nd[0] == lp.n_own + sum(lp.ring_counts[:L - 1]) and all(a >= b for a, b in zip(nd, nd[1:]))
        gl = np.concatenate([lp.own_gid, lp.halo_gid])
        assert len(np.unique(gl)) == len(gl)
        # the in-edges of every computed cell, 4 each, grouped by destination in global edge order; sources of layer l's destinations lie inside
        # what layer l-1 computed (the input rows for layer 0)
        assert lp.edge_index.shape[1] == 4 * nd[0] and np.array_equal(lp.edge_index[1], np.repeat(np.arange(nd[0]), 4))
        assert np.all(np.diff(lp.edge_gid.reshape(-1, 4), axis=1) > 0)
        assert np.array_equal(gl[lp.edge_index[0]], ei[0][lp.edge_gid]) and np.array_equal(gl[lp.edge_index[1]], ei[1][lp.edge_gid])
        reach = [lp.n_own + lp.n_halo] + nd
        for l in range(L):
            assert lp.edge_index[0][:4 * nd[l]].max() < reach[l]
        # ring k is exactly k hops out
        off = lp.n_own
        dist_k = np.full(n, -1); dist_k[lp.own_gid] = 0
        for k in range(1, L + 1):
            nb = np.unique(ei[0][(dist_k[ei[1]] == k - 1) & (dist_k[ei[0]] < 0)])
            dist_k[nb] = k
            assert np.array_equal(nb, gl[off:off + lp.ring_counts[k - 1]])
            off += lp.ring_counts[k - 1]
        h = x[torch.from_numpy(gl)][:, 1:]
        ea_local = ea[torch.from_numpy(lp.edge_gid)]
        e_loc = torch.from_numpy(lp.edge_index)
        with torch.no_grad():
            for i in range(L):
                blk = net.convs[i]
                h = blk[2](blk[1](blk[0]((h, h[:nd[i]]), ea_local[:4 * nd[i]], e_loc[:, :4 * nd[i]])))
            got[lp.own_gid] = net.decoder(h).numpy()
    assert not np.isnan(got).any() and np.abs(got - ref).max() <= 1e-5


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    adj, cent, x, ea = _scene()
    ei = adj.T.astype(np.int64)
    lp = build_local_part(ei, rcb_partition(cent, world), rank, world)
    net = oracle_static()
    rows = np.concatenate([lp.own_gid, lp.halo_gid])
    x_local = x[torch.from_numpy(rows)][:, 1:]
    ea_local = ea[torch.from_numpy(lp.edge_gid)]
    e_loc = torch.from_numpy(lp.edge_index)
    exchange = HaloExchange(lp, "cpu")

    calls = []

    def layer_fn(i, h, out, b, e):
        # destinations [b, e) only: their in-edges are the local rows 4b .. 4e-1 (list grouped by destination)
        calls.append((i, b, e))
        blk = net.convs[i]
        ei_sub = e_loc[:, 4 * b:4 * e].clone()
        ei_sub[1] -= b
        out[b:e] = blk[2](blk[1](blk[0]((h, h[b:e]), ea_local[4 * b:4 * e], ei_sub)))

    with torch.no_grad():
        # NaN-poisoned buffers: a boundary row computed before its halo arrived would show up in the logits
        logits = run_partitioned_layers(lp, x_local, net.num_layers, layer_fn, net.decoder, exchange,
                                        lambda r, c: torch.full((r, c), float("nan")), widths=[64, 128, 128, 128])
    assert calls[0] == (0, 0, lp.n_own) and calls[1] == (1, 0, lp.n_interior) and calls[2] == (1, lp.n_interior, lp.n_own)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gid=lp.own_gid, logits=logits.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_oracle_forward_matches_single_process(world, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    adj, _, x, ea = _scene()
    n = adj.shape[0] // 4
    torch.set_num_threads(1)
    net = oracle_static()
    with torch.no_grad():
        ref = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=torch.from_numpy(adj.T.astype(np.int64)))).numpy()
    got = np.full_like(ref, np.nan)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        got[d["gid"]] = d["logits"]
    assert not np.isnan(got).any()
    # same per-destination order; BLAS may pick different kernels for different row counts -> tiny tolerance
    assert np.abs(got - ref).max() <= 1e-5


def _dp_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dgnn_amd.partition import allreduce_gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    g = torch.Generator().manual_seed(100 + rank)  # every rank sees its own shard
    x, y = torch.randn(32, 8, generator=g), torch.randn(32, 2, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    allreduce_gradients(net)
    torch.save([p.grad.clone() for p in net.parameters()], os.path.join(out_dir, "g%d.pt" % rank))
    dist.destroy_process_group()


def test_dp_gradient_allreduce(tmp_path):
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = (torch.load(os.path.join(str(tmp_path), "g%d.pt" % r)) for r in range(world))
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    # equals the mean of the per-shard gradients computed in one process
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    acc = [torch.zeros_like(p) for p in net.parameters()]
    for r in range(world):
        net.zero_grad()
        g = torch.Generator().manual_seed(100 + r)
        x, y = torch.randn(32, 8, generator=g), torch.randn(32, 2, generator=g)
        ((net(x) - y) ** 2).mean().backward()
        for a, p in zip(acc, net.parameters()):
            a += p.grad / world
    for a, b in zip(acc, g0):
        assert torch.allclose(a, b, atol=1e-7)


# ---- backward of a partitioned single scene (round 4; SURVEY 8e "Backward mirrors it") ----------------------------------------------------------
def _whole_graph_train_step(dtype=torch.float64):
    """the reference's train-mode forward on the WHOLE scene as one block per layer (SurfaceNet.forward :196-227), the Trainer's kl loss over every
    cell, backward: logits, loss, gradients, BatchNorm buffers of a single process"""
    adj, cent, x, ea = _scene()
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64))
    net = oracle_static(train=True, dtype=dtype)
    x, ea = x.to(dtype), ea.to(dtype)
    x[:, 0] = x[:, 0].abs() + 0.05
    occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
    y = torch.cat([occ, 1 - occ], 1)
    whole = [(ei, torch.arange(4 * n), (n, n))] * net.num_layers
    logits = net(Config(all=Config(x=x, edge_attr=ea), batch_n_id=torch.arange(n), batch_adjs=whole))
    import torch.nn.functional as F
    w = x[:, 0]
    loss = (F.kl_div(F.log_softmax(logits, dim=-1), y, reduction="none").sum(1) * w).sum() / w.sum()
    loss.backward()
    return net, logits.detach(), loss.detach(), x, ea, y, adj, cent


def _train_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from dgnn_amd.partition import allreduce_gradients, partitioned_kl_loss, partitioned_train_forward
    adj, cent, x, ea = _scene()
    dtype = torch.float64
    x, ea = x.to(dtype), ea.to(dtype)
    x[:, 0] = x[:, 0].abs() + 0.05
    occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
    y = torch.cat([occ, 1 - occ], 1)
    ei = adj.T.astype(np.int64)
    lp = build_local_part(ei, rcb_partition(cent, world), rank, world)
    rows = torch.from_numpy(np.concatenate([lp.own_gid, lp.halo_gid]))
    own = torch.from_numpy(lp.own_gid)
    net = oracle_static(train=True, dtype=dtype)
    exchange = HaloExchange(lp, "cpu")
    logits = partitioned_train_forward(lp, x[rows][:, 1:], ea[torch.from_numpy(lp.edge_gid)], torch.from_numpy(lp.edge_index),
                                       [blk[0] for blk in net.convs], [blk[1].module for blk in net.convs], net.decoder, exchange)
    loss = partitioned_kl_loss(logits, y[own], x[own, 0])
    loss.backward()
    allreduce_gradients(net, average=False)        # partial sums over the owned cells -> the scene's gradient on every rank
    torch.save(dict(gid=own, logits=logits.detach(), loss=loss.detach(), grads={k: p.grad.clone() for k, p in net.named_parameters()},
                    buffers={k: b.clone() for k, b in net.named_buffers()}), os.path.join(out_dir, "t%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_backward_matches_the_whole_graph_backward(world, tmp_path):
    """halo rows' gradients return to their owners, BatchNorm(train) statistics and their backward sums span every rank, parameter gradients are summed:
    logits, loss, EVERY parameter gradient and the BatchNorm running buffers of `world` processes equal the single-process whole-graph step (fp64)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_train_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    net, logits, loss, *_ = _whole_graph_train_step()
    outs = [torch.load(os.path.join(str(tmp_path), "t%d.pt" % r)) for r in range(world)]
    got = torch.full_like(logits, float("nan"))
    for o in outs:
        got[o["gid"]] = o["logits"]
    assert (got - logits).abs().max().item() <= 1e-10 * max(1.0, logits.abs().max().item())
    ref_g = {k: p.grad for k, p in net.named_parameters()}
    gmax = max(g.abs().max().item() for g in ref_g.values())
    for o in outs:
        assert abs(o["loss"].item() - loss.item()) <= 1e-12 * max(1.0, abs(loss.item()))
        for k, g in o["grads"].items():
            assert (g - ref_g[k]).abs().max().item() <= 1e-7 * gmax, k        # (fp64; the scene-wide variance is E[x^2] - mean^2 of all-reduced sums, torch's is two-pass)
        for k, b in dict(net.named_buffers()).items():
            assert (o["buffers"][k].double() - b.double()).abs().max().item() <= 1e-7 * max(1.0, b.double().abs().max().item()), k
