"""Counterpart of the reference's learning/runModel.py for the part that drives the hot path:
``Trainer.train`` (:264-282), ``Trainer.train_test`` (:285-405: epochs, step LR decay, periodic validation,
``model_best.ptm`` / ``model_<epoch>.ptm``), ``Trainer.inference`` (:412-451), ``calcLossAndOA`` (:163-259, cell loss),
``calcRegularization`` (:109-160), ``Metrics`` (:48-80), ``adjust_learning_rate`` (:95-99) and the resume step of
run.py:102-113 (``load_epoch``).

Same names, arguments and error convention (``print`` + ``sys.exit(1)`` for config errors).  The model calls run
on the HIP kernels; the loss itself is a few elementwise ops on [batch, 2] tensors and uses torch on the GPU.
New (no reference counterpart, single process there): ``group=`` on ``train`` / ``train_test`` = data-parallel
replicas, one flat RCCL all-reduce of the gradients between ``backward()`` and ``step()`` (BASELINE config 5).
Validation by mesh metrics (chamfer / iou, :355-362) needs the reference's CPU post-processing (gco, trimesh) and is used
only when those import; the loss metric always works.  The device string is no longer hard-wired to cuda:<gpu> (:287).
"""
from __future__ import annotations

import csv
import os
import sys
from datetime import datetime
from shutil import copyfile

import torch
import torch.nn.functional as F

from .. import functional as Fn

# the "kl" cell loss (+ OA counter) as one library call each way on GPU logits; DGNN_FUSED_LOSS=0 runs the reference's op chain
FUSED_KL_LOSS = os.environ.get("DGNN_FUSED_LOSS", "1") != "0"


class Metrics:
    """Running sums, reference :48-80.  The reference reads every item back to the host as it is added (`.item()`, three
    device round trips per training step); here tensor items are added ON THEIR DEVICE into fp64 / int64 accumulators -- the
    same double-precision sums -- and come back to the host only when a sum is read (the print / validation cadence), so the
    training step enqueues without draining the GPU."""

    _FIELDS = ("samples_sum", "OA_sum", "cell_sum", "weight_sum", "reg_sum", "edges_sum")

    def __init__(self):
        self._host = dict.fromkeys(self._FIELDS, 0)
        self._dev = {}

    def _add(self, name, v):
        if isinstance(v, torch.Tensor) and v.is_cuda:
            v = v.detach().reshape(())
            acc = self._dev.get(name)
            if acc is None or acc.device != v.device:
                if acc is not None:
                    self._host[name] += acc.item()
                self._dev[name] = v.to(torch.float64 if v.is_floating_point() else torch.int64, copy=True)
            else:
                acc.add_(v)
        else:
            self._host[name] += v.item() if isinstance(v, torch.Tensor) else v

    def __getattr__(self, name):
        if name in Metrics._FIELDS:
            if name in Metrics._PACKED:
                self._flush_packed()
            acc = self.__dict__.get("_dev", {}).get(name)
            return self._host[name] + (acc.item() if acc is not None else 0)
        raise AttributeError(name)

    def addPacked(self, sums, samples):
        """`sums` = device fp64 [cell_sum, weight_sum, OA_sum] of one batch (the fused loss kernel's by-product): one add"""
        acc = self._dev.get("_packed")
        if acc is None or acc.device != sums.device:
            if acc is not None:
                self._flush_packed()
            self._dev["_packed"] = sums.detach().clone()
        else:
            acc.add_(sums)
        self._host["samples_sum"] += samples

    def packedAccumulator(self, device, samples):
        """the device fp64 [cell_sum, weight_sum, OA_sum] accumulator itself, for a kernel that adds a batch's sums into it (ops.kl_cell_loss_step:
        the loss launch does addPacked's add); `samples` rows are counted here"""
        acc = self._dev.get("_packed")
        if acc is None or acc.device != torch.device(device):
            if acc is not None:
                self._flush_packed()
            acc = self._dev["_packed"] = torch.zeros(3, dtype=torch.float64, device=device)
        self._host["samples_sum"] += samples
        return acc

    _PACKED = ("cell_sum", "weight_sum", "OA_sum")

    def _flush_packed(self):
        acc = self._dev.pop("_packed", None)
        if acc is not None:
            for name, v in zip(self._PACKED, acc.tolist()):
                self._host[name] += int(v) if name == "OA_sum" else v

    def addOAItem(self, oa, samples):
        self._add("OA_sum", oa)
        self._add("samples_sum", samples)

    def addCellLossItem(self, cell_loss, weight):
        self._add("cell_sum", cell_loss)
        self._add("weight_sum", weight)

    def addRegLossItem(self, reg_loss, edges):
        self._add("reg_sum", reg_loss)
        self._add("edges_sum", edges)

    def getOA(self):
        return self.OA_sum * 100 / max(self.samples_sum, 1)

    def getCellLoss(self):
        w = self.weight_sum
        return self.cell_sum / w if w else 0.0

    def getRegLoss(self):
        e = self.edges_sum
        return self.reg_sum / e if e else 0.0


TRAIN_DIRECT = __import__("os").environ.get("DGNN_TRAIN_DIRECT", "1") != "0"
KL_LOSS_ONE_LAUNCH = __import__("os").environ.get("DGNN_KL_LOSS_ONE_LAUNCH", "1") != "0"


def make_adam(params, lr):
    """the reference's optimizer (:290 torch.optim.Adam(model.parameters(), lr)); fp32 parameters on a GPU are stepped by one library launch
    (dgnn_amd.optim.Adam: same rule and state), anything else -- and DGNN_TORCH_ADAM=1 -- by torch's own"""
    import os
    from ..optim import Adam
    params = list(params)
    if os.environ.get("DGNN_TORCH_ADAM") != "1" and params and Adam.supports(params):
        return Adam(params, lr=lr)
    return torch.optim.Adam(params, lr=lr)


def adjust_learning_rate(optimizer, clf):
    """lr * 0.1 ** (epoch // adjust_lr_every), reference :95-99."""
    lr = clf.training.learning_rate * (0.1 ** (clf.temp.current_epoch // clf.training.adjust_lr_every))
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


class Trainer:

    def __init__(self, model):
        self.model = model

    def calcRegularization(self, logits_cell, data, clf, metrics):
        """Edge total-variation term on inside-probabilities, reference :109-160."""
        if data.batch_adjs:
            adj = data.batch_adjs[self.model.num_layers]
            inner = F.softmax(logits_cell[:adj.size[0]], dim=-1)
            ei = adj.edge_index.to(logits_cell.device)
        else:
            inner = F.softmax(logits_cell, dim=-1)
            ei = data.edge_index.to(logits_cell.device)
        tv = torch.abs(inner[ei[0, :]][:, 0] - inner[ei[1, :]][:, 0])
        reg_loss = tv * clf.regularization.edge_weight
        metrics.addRegLossItem(reg_loss.sum(), tv.size(0))
        return reg_loss.mean()

    def calcLossAndOA(self, logits_cell, logits_edge, data, clf, metrics):
        """Volume-weighted cell loss, reference :163-259 (kl / bce / mse)."""
        dev = logits_cell.device
        if not clf.regularization.cell_type:
            # the reference computes the cell loss only under `if (clf.regularization.cell_type)` (:169) and then reads the
            # unassigned `cell_loss` (:223): same exception type, with the reason spelled out
            raise UnboundLocalError("local variable 'cell_loss' referenced before assignment: the reference's calcLossAndOA "
                                    "needs regularization.cell_type (learning/runModel.py:169,223)")
        gt = data.batch_gt.to(dev)
        if clf.training.loss == "kl" and FUSED_KL_LOSS and logits_cell.is_cuda and logits_cell.dtype == torch.float32 and logits_cell.size(0) > 0 \
                and clf.regularization.cell_norm in Fn.ops.CELL_NORMS and gt.dtype == torch.float32:
            # log_softmax, kl_div, the volume weights, both sums, the quotient and the OA counter in one launch (and one backward)
            vol = data.batch_x[:, 0].to(dev)
            loss, sums = Fn.kl_cell_loss(logits_cell, gt, vol if vol.dtype == torch.float32 else vol.float(), clf.regularization.cell_norm)
            metrics.addPacked(sums, data.batch_x.shape[0])
            return self._with_regularization(loss, logits_cell, data, clf, metrics)
        if clf.training.loss == "kl":
            cell_loss = F.kl_div(F.log_softmax(logits_cell, dim=-1), gt[:, :2], reduction='none').sum(dim=1)
            pred = logits_cell.argmax(1)
            metrics.addOAItem(((gt[:, 0] > gt[:, 1]).long() == pred).sum(), data.batch_x.shape[0])
        elif clf.training.loss == "bce":
            cell_loss = F.binary_cross_entropy_with_logits(logits_cell.squeeze(-1), gt[:, 3], reduction='none')
            metrics.addOAItem((gt[:, 3] == torch.round(torch.sigmoid(logits_cell.squeeze(-1)))).sum(), data.batch_x.shape[0])
        elif clf.training.loss == "mse":
            cell_loss = F.mse_loss(torch.sigmoid(logits_cell).squeeze(), gt[:, 0])
        else:
            print("{} is not a valid loss. choose either kl or mse".format(clf.training.loss))
            sys.exit(1)
        vol = data.batch_x[:, 0].to(dev)
        if clf.regularization.cell_norm == "log":
            w = torch.log(1 + vol)
        elif clf.regularization.cell_norm == "sqrt":
            w = torch.sqrt(vol)
        else:
            w = vol
        cell_loss = cell_loss * w
        cell_sum, w_sum = cell_loss.sum(), w.sum()
        metrics.addCellLossItem(cell_sum, w_sum)
        return self._with_regularization(cell_sum / w_sum, logits_cell, data, clf, metrics)

    def _with_regularization(self, loss, logits_cell, data, clf, metrics):
        if clf.regularization.edge_epoch is not None:
            if clf.graph.additional_num_hops != 1:
                print("ERROR: clf.graph.additional_num_hops has to be >= 1 to use regularization")
                sys.exit(1)
            if clf.temp.current_epoch >= clf.regularization.edge_epoch:
                loss = loss + self.calcRegularization(logits_cell, data, clf, metrics)
        return loss

    def _all_training(self):
        """model.train() (reference :266) writes the flag of ~40 modules through nn.Module.__setattr__ on every step; when every flag is
        already set it has nothing to do -- reading them costs a fifth of that."""
        mods = self.__dict__.get("_mods")
        if mods is None or mods[0] is not self.model:
            mods = self.__dict__["_mods"] = (self.model, list(self.model.modules()))      # (modules() walks the tree: 40 us a step)
        for m in mods[1]:
            if not m.training:
                return False
        return True

    def train(self, data_train, optimizer, clf, group=None):
        """One optimisation step on one sampled batch, reference :264-282.  `group`: data-parallel replicas (one scene shard
        per rank): the flat gradient is all-reduced (mean) between backward() and step(); None + an initialised default
        process group of size > 1 uses that group, a single process changes nothing."""
        if not self._all_training():
            self.model.train()
        direct = self._train_direct(data_train, optimizer, clf, group)
        if direct is not None:
            return direct
        logits_cell = self.model(data_train)
        n_sup = data_train.batch_adjs[self.model.num_layers - 1].size[1] if hasattr(data_train.batch_adjs[0], "size") \
            else data_train.batch_adjs[self.model.num_layers - 1][2][1]
        data_train.batch_x, data_train.batch_gt = self._batch_rows(data_train, n_sup)
        loss = self.calcLossAndOA(logits_cell, None, data_train, clf, clf.training.metrics)
        optimizer.zero_grad()
        loss.backward()
        from ..partition import allreduce_gradients
        allreduce_gradients(self.model, group)   # no-op without a process group / with one rank
        optimizer.step()
        return loss.detach()

    @staticmethod
    def _batch_rows(data_train, n_sup):
        """all.x[ids], all.y[ids] for the batch's targets (reference :273-274): from the block's builder when the loader gathered them behind the block
        (attach_block_rows), else indexed here"""
        from ..sampler import block_rows
        x_all, y_all, n_id = data_train.all.x, data_train.all.y, data_train.batch_n_id
        bx = block_rows(n_id, x_all, 0, x_all.size(1), "batch") if x_all.dim() == 2 else None
        by = block_rows(n_id, y_all, 0, y_all.size(1), "batch") if y_all.dim() == 2 else None
        if bx is not None and by is not None and bx.size(0) == n_sup and by.size(0) == n_sup:
            return bx, by
        ids = n_id[:n_sup].to(x_all.device)
        return x_all[ids], y_all[ids]

    @staticmethod
    def attach_block_rows(loader, data_all, model):
        """Tells a dgnn_amd NeighborSampler to gather, behind every block and on its own stream, the rows a training step indexes at its head: the
        model's input rows x[n_id, 1:] (or x[n_id]) and the targets' x / y rows.  Any other loader: nothing happens, the step indexes as before."""
        attach = getattr(loader, "attach_rows", None)
        x_all, y_all = getattr(data_all, "x", None), getattr(data_all, "y", None)
        if attach is None or not isinstance(x_all, torch.Tensor) or x_all.dim() != 2:
            return
        clf = getattr(model, "clf", None)
        drop = False
        if clf is not None:
            reg, feat = getattr(clf, "regularization", None), getattr(clf, "features", None)
            if type(model).__module__.endswith("UpdatedEdgeFilters"):
                drop = bool(feat is not None and feat.normalization_feature and not feat.keep_normalization_feature)
            else:
                drop = bool(reg is not None and reg.cell_type)
        col0 = 1 if drop else 0
        specs = [(x_all, col0, x_all.size(1) - col0, "all"), (x_all, 0, x_all.size(1), "batch")]
        if isinstance(y_all, torch.Tensor) and y_all.dim() == 2:
            specs.append((y_all, 0, y_all.size(1), "batch"))
        attach(specs)

    def _train_direct(self, data_train, optimizer, clf, group):
        """The step without the autograd engine (DGNN_TRAIN_DIRECT=0 keeps the autograd path): the Static model's whole-model library calls, the
        fused kl loss and its gradient issued directly (SurfaceNet.train_step_direct) -- the step is bound by the host's issue rate, and the engine's
        bookkeeping, the loss Function and zero_grad were a fifth of it.  Same kernels in the same order: same numbers as the autograd path
        (tests/test_gpu_train.py).  None = this configuration takes the autograd path (other losses / models, an active edge regulariser)."""
        model = self.model
        step = getattr(model, "train_step_direct", None)
        if step is None or not TRAIN_DIRECT or clf.training.loss != "kl" or not FUSED_KL_LOSS or not clf.regularization.cell_type \
                or clf.regularization.cell_norm not in Fn.ops.CELL_NORMS:
            return None
        if clf.regularization.edge_epoch is not None and (clf.temp.current_epoch >= clf.regularization.edge_epoch or clf.graph.additional_num_hops != 1):
            return None        # (the second clause: `_with_regularization` prints and exits for such a config BEFORE edge_epoch is reached -- reference :250-253)
        x_all, y_all = data_train.all.x, data_train.all.y
        if not (x_all.is_cuda and y_all.is_cuda and x_all.dtype == torch.float32 and y_all.dtype == torch.float32):
            return None
        n_sup = data_train.batch_adjs[model.num_layers - 1].size[1] if hasattr(data_train.batch_adjs[0], "size") \
            else data_train.batch_adjs[model.num_layers - 1][2][1]
        if n_sup == 0:
            return None        # (the fused kl loss is guarded by logits.size(0) > 0 on the autograd path; an empty batch takes that path)
        data_train.batch_x, data_train.batch_gt = self._batch_rows(data_train, n_sup)     # (the reference leaves these on the data object, :273-274)
        metrics, norm = clf.training.metrics, Fn.ops.CELL_NORMS[clf.regularization.cell_norm]
        one = self.__dict__.get("_one")
        if one is None or one.device != x_all.device:
            one = self._one = torch.ones((), dtype=torch.float32, device=x_all.device)

        def loss_fn(logits):
            if logits.dim() != 2 or logits.size(1) != 2 or logits.size(0) != n_sup:
                raise RuntimeError("train: the model returned %s logits for %d targets" % (tuple(logits.shape), n_sup))
            vol = data_train.batch_x[:, 0]
            if KL_LOSS_ONE_LAUNCH and hasattr(metrics, "packedAccumulator"):
                # forward, the metric sums' accumulation and the gradient in one launch (round 6; same bits as the calls below)
                got = Fn.ops.kl_cell_loss_step(logits, data_train.batch_gt, vol, norm, running=metrics.packedAccumulator(logits.device, 0))
                if got is not None:
                    metrics.packedAccumulator(logits.device, n_sup)
                    return got[0], got[2]
            loss, sums = Fn.ops.kl_cell_loss_fwd(logits, data_train.batch_gt, vol, norm)
            metrics.addPacked(sums, n_sup)
            return loss, Fn.ops.kl_cell_loss_bwd(logits, data_train.batch_gt, vol, norm, sums, one)
        loss = step(data_train, loss_fn)
        if loss is None:
            return None
        from ..partition import allreduce_gradients
        allreduce_gradients(model, group)
        optimizer.step()
        return loss

    def train_test(self, data, clf, group=None):
        """Epoch loop of the reference (:285-405): Adam, lr * 0.1^(epoch // adjust_lr_every), one `train` per sampled batch,
        running metrics printed every `print_every` iterations, validation every `val_every` iterations (loss / OA over
        data.validation.all through `inference`; chamfer / iou only when the reference's mesh post-processing imports),
        `models/model_best.ptm` whenever the validation metric improves and `models/model_<epoch>.ptm` every `export_every`
        iterations -- plain state_dicts, exactly what run.py:148-157 / :102-113 load.  Rows of the results table go to
        clf.files.results (csv) when that is configured.  With `group` (data-parallel replicas) only rank 0 writes; every epoch runs as many
        steps as the rank with the fewest batches has (`_agreed_steps`); parameters are identical on all ranks after every step, BatchNorm
        running statistics are NOT synchronised (each rank's follow its own shard, as a single process's follow its scene) and the
        checkpoints hold rank 0's."""
        if not getattr(clf.temp, "device", None):
            clf.temp.device = "cuda:" + str(clf.temp.args.gpu)
        from ..partition import broadcast_parameters
        broadcast_parameters(self.model, group)   # data-parallel replicas start equal (no-op for a single process)
        optimizer = make_adam(self.model.parameters(), clf.training.learning_rate)
        import torch.distributed as dist
        writer = (not (dist.is_available() and dist.is_initialized())) or dist.get_rank(group) == 0
        models_dir = os.path.join(clf.paths.out, "models")
        if writer:
            os.makedirs(models_dir, exist_ok=True)
        results = getattr(getattr(clf, "files", None), "results", None) if hasattr(clf, "files") else None
        rows = []
        metric_name = clf.temp.metrics[0] if getattr(clf.temp, "metrics", None) else "loss"
        if not hasattr(clf, "best_metric") or clf.best_metric is None:
            clf.best_metric = float("-inf") if metric_name == "iou" else float("inf")
        clf.training.metrics = Metrics()
        iterations = 0
        row = {}
        self.attach_block_rows(data.train.batches, data.train.all, self.model)
        for current_epoch in range(1, clf.training.epochs + 1):
            clf.temp.current_epoch = current_epoch
            adjust_learning_rate(optimizer, clf)
            for data.train.batch_size, data.train.batch_n_id, data.train.batch_adjs in _agreed_steps(data.train.batches, group, clf.temp.device):
                iterations += 1
                self.train(data.train, optimizer, clf, group)
                printing = (iterations % clf.training.print_every) == 0 or iterations == 1
                validating = (iterations % clf.training.val_every) == 0 and getattr(data, "validation", None) is not None
                if printing or validating:
                    # the running sums live on the device (Metrics): they are read back only at the cadence at which they are consumed, so the
                    # iterations in between enqueue without draining the GPU (the block builder's prefetch then overlaps them)
                    m = clf.training.metrics
                    row.update(iteration=iterations, epoch=current_epoch, train_loss_cell=m.getCellLoss(), train_loss_reg=m.getRegLoss(),
                               train_loss_total=m.getRegLoss() + m.getCellLoss(), train_OA=m.getOA())
                if printing:
                    if writer:
                        print('%s[%3d] Epoch %3d -> Train Loss (cell): %1.4f,  Train Loss (reg): %1.4f, Train Loss (total): %1.4f,  Train OA: %3.2f%%'
                              % (datetime.now().strftime("[%H:%M:%S]"), iterations, current_epoch, row['train_loss_cell'],
                                 row['train_loss_reg'], row['train_loss_total'], row['train_OA']))
                    clf.training.metrics = Metrics()
                if (iterations % clf.training.val_every) == 0 and getattr(data, "validation", None) is not None:
                    OA = loss = reg = samples = weight = edges = 0
                    current_metric = 0
                    mesh_metric = metric_name in ("chamfer", "iou")
                    for i, d in enumerate(data.validation.all):
                        loader = data.validation.batches[i] if clf.validation.batch_size else []
                        prediction = self.inference(d, loader, clf)
                        im = clf.inference.metrics
                        OA += im.OA_sum
                        samples += im.samples_sum
                        reg += im.reg_sum
                        edges += im.edges_sum
                        loss += im.cell_sum
                        weight += im.weight_sum
                        if mesh_metric:
                            from ..processing import generate_mesh as gm
                            _, eval_dict = gm.generate(d, prediction, clf)   # raises without trimesh (as the reference does)
                            current_metric += eval_dict[metric_name]
                    re = reg / edges if (reg > 0.0 and edges > 0.0) else 0.0
                    loss /= weight
                    loss_total = re + loss
                    if metric_name == "loss":
                        current_metric = loss_total
                        improved = current_metric < clf.best_metric
                    else:
                        current_metric /= len(data.validation.all)
                        improved = current_metric > clf.best_metric if metric_name == "iou" else current_metric < clf.best_metric
                    if improved:
                        clf.best_metric = current_metric
                        if writer:
                            torch.save(self.model.state_dict(), os.path.join(models_dir, "model_best.ptm"))
                    row.update(test_loss_cell=loss, test_loss_reg=re, test_loss_total=loss_total, test_OA=OA * 100 / max(samples, 1))
                    row['test_current_' + metric_name] = current_metric
                    row['test_best_' + metric_name] = clf.best_metric
                    rows.append(dict(row))
                    if writer and results:
                        keys = sorted({k for r in rows for k in r})
                        with open(results, "w", newline="") as f:
                            w = csv.DictWriter(f, fieldnames=keys)
                            w.writeheader()
                            w.writerows(rows)
                if (iterations % clf.training.export_every) == 0 and writer:
                    model_path = os.path.join(models_dir, "model_" + str(int(current_epoch)) + ".ptm")
                    print('[{}] Epoch {} -> Export model to {}'.format(iterations, current_epoch, model_path))
                    torch.save(self.model.state_dict(), model_path)
                    if results and os.path.isfile(results):
                        copyfile(results, os.path.splitext(results)[0] + "_" + str(current_epoch) + ".csv")
        clf.results_rows = rows
        return rows

    def inference(self, data_inference, subgraph_loader, clf):
        """Dispatch of the three inference schedules, reference :412-451.  Logits come back on the CPU."""
        self.model.eval()
        with torch.no_grad():
            if clf.inference.per_layer and clf.temp.batch_size:
                assert subgraph_loader and len(subgraph_loader.sizes) == 1
                logits_cell = self.model.inference_layer_batch(data_inference, subgraph_loader)
            elif clf.inference.per_layer and not clf.temp.batch_size:
                assert not subgraph_loader
                logits_cell = self.model.inference_layer(data_inference)
            elif not clf.inference.per_layer and clf.temp.batch_size:
                assert subgraph_loader
                logits_cell = self.model.inference_batch_layer(data_inference, subgraph_loader)
            else:
                print("not a valid inference method set either per_layer to true or specify batch_size")
                sys.exit(1)
        if clf.inference.has_label:
            clf.inference.metrics = Metrics()
            data_inference.batch_x = data_inference.x
            data_inference.batch_gt = data_inference.y
            data_inference.batch_adjs = []
            self.calcLossAndOA(logits_cell, None, data_inference, clf, clf.inference.metrics)
        if clf.training.loss == "mse":
            logits_cell = torch.cat((1 - logits_cell, logits_cell), dim=1)
        return logits_cell.to('cpu')


def _agreed_steps(batches, group=None, device=None):
    """Iterates `batches`, but with data-parallel replicas (an initialised process group of more than one rank) only for as many steps as
    EVERY rank has: each step holds one gradient all-reduce, so a rank whose shard yields more batches than another's would wait in a
    collective nobody else enters.  With a length the ranks agree on min(len) once per epoch; without one they agree step by step on
    whether all of them still have a batch (a 1-element all-reduce).  Surplus batches of the longer shards are dropped -- what
    DistributedSampler(drop_last=True) does.  A single process iterates `batches` unchanged."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        yield from batches
        return
    on_gpu = dist.get_backend(group) == "nccl"     # RCCL reduces device tensors only
    # ... on the TRAINER's GPU (clf.temp.device = "cuda:<n>", addressed without set_device as the reference does): a bare "cuda" is the thread's
    # current device, which RCCL would take for a device mismatch
    where = (device if (device is not None and str(device).startswith("cuda")) else "cuda") if on_gpu else "cpu"
    if hasattr(batches, "__len__"):
        n = torch.tensor([len(batches)], dtype=torch.int64, device=where)
        dist.all_reduce(n, op=dist.ReduceOp.MIN, group=group)
        steps = int(n.item())
        it = iter(batches)
        for _ in range(steps):
            yield next(it)
        close = getattr(it, "close", None)
        if close is not None:
            close()
        return
    it = iter(batches)
    while True:
        item = next(it, None)
        have = torch.tensor([0 if item is None else 1], dtype=torch.int64, device=where)
        dist.all_reduce(have, op=dist.ReduceOp.MIN, group=group)
        if int(have.item()) == 0:
            return
        yield item


def load_epoch(model, clf):
    """Resume step of run.py:102-113: training.load_epoch names `<out>/models/model_<load_epoch>.ptm`, a plain state_dict
    (optimizer state, epoch counter and RNG are not part of the reference's checkpoints).  Returns True when loaded."""
    if not clf.training.load_epoch:
        return False
    model_file = os.path.join(clf.paths.out, "models", "model_" + str(clf.training.load_epoch) + ".ptm")
    print("\nLoad existing model at epoch ", clf.training.load_epoch)
    if not os.path.isfile(model_file):
        print("\nERROR: The model {} does not exist. Check that you have set the correct path in data:out in the config file!".format(model_file))
        sys.exit(1)
    model.load_state_dict(torch.load(model_file, map_location="cpu"))
    return True
