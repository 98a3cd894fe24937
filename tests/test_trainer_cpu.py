"""CPU tests of the runModel.py counterpart and of the data-parallel wiring (BASELINE config 5).

The HIP SurfaceNet has no CPU path, so here ``Trainer`` drives the ORACLE's SurfaceNet (same module surface: ``model(data)``,
``num_layers``, ``inference_layer`` ...): what is tested is the host logic -- epoch loop, learning-rate schedule, checkpoint
names, resume, the gradient all-reduce between backward() and step() -- not the kernels.  The GPU twins of these tests
(tests/test_gpu_train.py) run the same flows on the HIP model."""
import os
import socket

import numpy as np
import pytest
import torch

from dgnn_amd.config import Config, load_config
from helpers import kf96_state_dict, oracle_static

HERE = os.path.dirname(os.path.abspath(__file__))


def small_scene(points=220, seed=3):
    from dgnn_amd.synthetic import delaunay_tet_graph
    adj, _, _ = delaunay_tet_graph(points, seed=seed)
    n = adj.shape[0] // 4
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 29, generator=g)
    x[:, 0] = x[:, 0].abs() + 0.05
    ea = torch.randn(4 * n, 20, generator=g)
    occ = torch.sigmoid(2 * x[:, 3:4] + x[:, 7:8])
    return adj, n, x, ea, torch.cat([occ, 1 - occ], 1)


def blocks(adj, n, targets, hops=4):
    from oracle.pyg_semantics import neighbor_sampler_full
    n_id, adjs = neighbor_sampler_full(adj.T.astype(np.int64), n, np.asarray(targets), hops)
    return torch.from_numpy(n_id), [(torch.from_numpy(a), torch.from_numpy(e), s) for a, e, s in adjs]


def make_clf(tmp_path=None):
    clf = load_config(os.path.join(HERE, "configs", "pretrained_like.yaml"))
    clf.temp.device = "cpu"
    clf.temp.num_node_features, clf.temp.num_edge_features = 28, 20
    clf.temp.current_epoch = 0
    clf.temp.batch_size = clf.inference.batch_size
    clf.temp.metrics = ["loss"]
    if tmp_path is not None:
        clf.paths.out = str(tmp_path)
        clf.files = Config(results=os.path.join(str(tmp_path), "results.csv"))
    return clf


def test_load_config_yaml_plumbing_builds_the_model_and_loads_the_checkpoint():
    """configs/pretrained/reconbench.yaml-style file -> Config -> SurfaceNet(clf): state_dict keys of the shipped
    checkpoint load strictly; a CPU device is refused loudly (no fallback)."""
    from dgnn_amd.learning.runModel import Trainer
    from dgnn_amd.learning.surfaceNetStaticEdgeFilters import SurfaceNet
    clf = make_clf()
    assert clf.model.convs == [64, 128, 128, 128] and clf.regularization.edge_type is None and clf.training.loss == "kl"
    assert clf.graph.clique_sizes == [-1] and clf.inference.per_layer == 1
    net = SurfaceNet(clf)
    missing = net.load_state_dict(kf96_state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert net.num_layers == 4 and sum(v.numel() for v in net.state_dict().values()) == 103699  # 49 tensors
    adj, n, x, ea, y = small_scene(60)
    data = Config(x=x, y=y, edge_attr=ea, edge_index=torch.from_numpy(adj.T.astype(np.int64)))
    with pytest.raises(RuntimeError, match="GPU only"):
        Trainer(net).inference(data, [], clf)


def test_train_test_loop_checkpoints_and_resume(tmp_path):
    """epochs x batches, lr decay, model_<epoch>.ptm / model_best.ptm, results csv; save -> load_epoch -> identical logits."""
    from dgnn_amd.learning.runModel import Trainer, load_epoch
    clf = make_clf(tmp_path)
    adj, n, x, ea, y = small_scene()
    ei = torch.from_numpy(adj.T.astype(np.int64))
    all_ = Config(x=x, y=y, edge_attr=ea)
    batches = [(32,) + blocks(adj, n, range(s, s + 32)) for s in (0, 32, 64)]
    val = Config(x=x, y=y, edge_attr=ea, edge_index=ei, infinite=torch.zeros(n))
    data = Config(train=Config(all=all_, batches=batches), validation=Config(all=[val], batches=[[]]))
    net = oracle_static(train=True)
    lrs = []
    import dgnn_amd.learning.runModel as rm
    orig = rm.adjust_learning_rate

    def spy(opt, c):
        orig(opt, c)
        lrs.append(opt.param_groups[0]["lr"])
    rm.adjust_learning_rate = spy
    try:
        rows = Trainer(net).train_test(data, clf)
    finally:
        rm.adjust_learning_rate = orig
    assert np.allclose(lrs, [0.005, 0.0005, 0.0005])            # epochs 1,2,3 with adjust_lr_every = 2 (:95-99)
    assert len(rows) == 4 and rows[-1]["iteration"] == 8          # 9 iterations, val_every 2
    assert rows[-1]["test_best_loss"] <= rows[0]["test_best_loss"]
    models = sorted(os.listdir(os.path.join(str(tmp_path), "models")))
    assert models == ["model_1.ptm", "model_2.ptm", "model_3.ptm", "model_best.ptm"]  # export_every 3 -> iterations 3, 6, 9
    assert os.path.isfile(clf.files.results)
    # resume: a fresh model + training.load_epoch reproduces the saved model's logits bit for bit
    net.eval()
    with torch.no_grad():
        want = net.inference_layer(val)
    clf.training.load_epoch = "3"
    net2 = oracle_static(load=False, seed=9)
    assert load_epoch(net2, clf)
    net2.eval()
    with torch.no_grad():
        assert torch.equal(net2.inference_layer(val), want)
    clf.training.load_epoch = "7"
    with pytest.raises(SystemExit):
        load_epoch(net2, clf)


def _dp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dgnn_amd.learning.runModel import Metrics, Trainer
    torch.set_num_threads(1)
    clf = make_clf()
    clf.training.metrics = Metrics()
    adj, n, x, ea, y = small_scene()
    net = oracle_static(train=True)
    opt = torch.optim.Adam(net.parameters(), lr=0.005)
    tr = Trainer(net)
    for step in range(2):
        n_id, adjs = blocks(adj, n, range(40 * rank + 8 * step, 40 * rank + 8 * step + 24))   # each rank its own shard
        tr.train(Config(all=Config(x=x, y=y, edge_attr=ea), batch_n_id=n_id, batch_adjs=adjs), opt, clf)
    torch.save({k: v for k, v in net.state_dict().items()}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_training_two_ranks_equals_mean_gradient_step(tmp_path):
    """World-2 gloo: Trainer.train all-reduces the gradients -> replicas' parameters stay bit-equal and equal a single
    process that averages the two shards' gradients by hand (BatchNorm running statistics stay per rank)."""
    import torch.multiprocessing as mp
    from dgnn_amd.learning.runModel import Metrics, Trainer
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sd = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(2)]
    pnames = [k for k, _ in oracle_static().named_parameters()]
    for k in pnames:
        assert torch.equal(sd[0][k], sd[1][k]), k
    assert not torch.equal(sd[0]["convs.0.norm.module.running_mean"], sd[1]["convs.0.norm.module.running_mean"])
    # single process, mean of the two shards' gradients
    torch.set_num_threads(1)
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
            nets[r].train()
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
        assert torch.equal(p.detach(), sd[0][k]), k


def _dp_unequal_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dgnn_amd.learning.runModel import Trainer
    torch.set_num_threads(1)
    clf = make_clf(os.path.join(out_dir, "r%d" % rank))
    clf.training.epochs, clf.training.val_every, clf.training.export_every = 2, 1000, 1000
    adj, n, x, ea, y = small_scene()
    # shards of different size: rank 0 has 4 batches per epoch, rank 1 only 2 -- as a list (has a length) or as a generator (has none)
    starts = (0, 24, 48, 72) if rank == 0 else (100, 124)
    batches = [(24,) + blocks(adj, n, range(s, s + 24)) for s in starts]

    class NoLen:
        def __iter__(self):
            return iter(batches)
    data = Config(train=Config(all=Config(x=x, y=y, edge_attr=ea), batches=batches if "list" in out_dir else NoLen()))
    net = oracle_static(train=True, load=False, seed=rank)      # replicas start different: train_test broadcasts rank 0's weights
    steps = []
    tr = Trainer(net)
    orig = tr.train
    tr.train = lambda *a, **k: (steps.append(1), orig(*a, **k))[1]
    tr.train_test(data, clf)
    torch.save({"sd": net.state_dict(), "steps": len(steps)}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["list", "generator"])
def test_dp_train_test_with_unequal_shards_runs_the_same_number_of_steps_on_every_rank(tmp_path, kind):
    """Two ranks whose shards hold 4 and 2 batches per epoch: every epoch runs min = 2 all-reduced steps on BOTH ranks (no rank is left
    alone in a collective), for batch sources with and without a length; the replicas' parameters end bit-equal."""
    import torch.multiprocessing as mp
    out = os.path.join(str(tmp_path), kind)
    os.makedirs(out)
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    mp.spawn(_dp_unequal_worker, args=(2, port, out), nprocs=2, join=True)
    r = [torch.load(os.path.join(out, "r%d.pt" % k)) for k in range(2)]
    assert r[0]["steps"] == r[1]["steps"] == 4          # 2 epochs x min(4, 2)
    for k, _ in oracle_static().named_parameters():
        assert torch.equal(r[0]["sd"][k], r[1]["sd"][k]), k


def test_batch_norm_momentum_none_is_the_cumulative_average():
    """torch.nn.BatchNorm1d(momentum=None): running statistics = cumulative average over the batches seen (factor 1 / num_batches_tracked);
    the composite training entry points decline such a layer (they take a fixed factor) and batch_norm_act computes the factor per call."""
    from dgnn_amd import functional as Fn
    bn = torch.nn.BatchNorm1d(8, momentum=None)
    assert not Fn.sage_train_layer_supported(torch.zeros(4, 8), None, bn) or not torch.cuda.is_available()
    import inspect
    src = inspect.getsource(Fn.batch_norm_act)
    assert "1.0 / float(bn.num_batches_tracked)" in src and "0.1" not in src


def test_metrics_running_sums_equal_the_reference_formulas():
    """Metrics (reference learning/runModel.py:48-80): OA = 100 * sum(correct) / sum(samples), cell loss = sum(cell) / sum(weight),
    reg loss = sum(reg) / sum(edges); items may be Python numbers or tensors (tensors on a GPU are summed there, see the class
    docstring -- here they are CPU tensors, the read-back path), and the packed form the fused loss kernel produces adds the same sums."""
    import torch
    from dgnn_amd.learning.runModel import Metrics
    m, p = Metrics(), Metrics()
    oa, n, cell, w = [120, 99, 250], [128, 128, 256], [0.5, 0.25, 1.125], [2.0, 1.5, 4.0]
    for a, b, c, d in zip(oa, n, cell, w):
        m.addOAItem(torch.tensor(a), b)
        m.addCellLossItem(torch.tensor(c), torch.tensor(d))
        p.addPacked(torch.tensor([c, d, float(a)], dtype=torch.float64), b)
    m.addRegLossItem(torch.tensor(3.0), 10)
    assert m.getOA() == p.getOA() == 100 * sum(oa) / sum(n)
    assert m.getCellLoss() == p.getCellLoss() == sum(cell) / sum(w)
    assert m.getRegLoss() == 0.3 and p.getRegLoss() == 0.0
    assert m.samples_sum == sum(n) and p.OA_sum == sum(oa) and Metrics().getOA() == 0 and Metrics().getCellLoss() == 0.0
