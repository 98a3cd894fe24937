"""Counterpart of the reference's learning/runModel.py for the part that drives the hot path:
``Trainer.train`` (:264-282), ``Trainer.inference`` (:412-451), ``calcLossAndOA`` (:163-259, cell loss) and
``calcRegularization`` (:109-160), ``Metrics`` (:48-80), ``adjust_learning_rate`` (:95-99).

Same names, arguments and error convention (``print`` + ``sys.exit(1)`` for config errors).  The model calls run
on the HIP kernels; the loss itself is a few elementwise ops on [batch, 2] tensors and uses torch on the GPU.
Validation-time mesh extraction / metrics (train_test :285-405) need the reference's CPU post-processing
(gco, trimesh) and stay out of scope; the device string is no longer hard-wired to cuda:<gpu> (:287).
"""
from __future__ import annotations

import sys

import torch
import torch.nn.functional as F


class Metrics:
    """Running sums, reference :48-80."""

    def __init__(self):
        self.samples_sum = 0
        self.OA_sum = 0
        self.cell_sum = 0
        self.weight_sum = 0
        self.reg_sum = 0
        self.edges_sum = 0

    def addOAItem(self, oa, samples):
        self.OA_sum += oa
        self.samples_sum += samples

    def addCellLossItem(self, cell_loss, weight):
        self.cell_sum += float(cell_loss)
        self.weight_sum += float(weight)

    def addRegLossItem(self, reg_loss, edges):
        self.reg_sum += float(reg_loss)
        self.edges_sum += edges

    def getOA(self):
        return self.OA_sum * 100 / max(self.samples_sum, 1)

    def getCellLoss(self):
        return self.cell_sum / self.weight_sum if self.weight_sum else 0.0

    def getRegLoss(self):
        return self.reg_sum / self.edges_sum if self.edges_sum else 0.0


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
        gt = data.batch_gt.to(dev)
        if clf.training.loss == "kl":
            cell_loss = F.kl_div(F.log_softmax(logits_cell, dim=-1), gt[:, :2], reduction='none').sum(dim=1)
            pred = logits_cell.argmax(1)
            metrics.addOAItem(int(((gt[:, 0] > gt[:, 1]).long() == pred).sum()), data.batch_x.shape[0])
        elif clf.training.loss == "bce":
            cell_loss = F.binary_cross_entropy_with_logits(logits_cell.squeeze(-1), gt[:, 3], reduction='none')
            metrics.addOAItem(int((gt[:, 3] == torch.round(torch.sigmoid(logits_cell.squeeze(-1)))).sum()), data.batch_x.shape[0])
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
        elif clf.regularization.cell_type:
            w = vol
        else:
            w = torch.ones_like(cell_loss)
        cell_loss = cell_loss * w
        metrics.addCellLossItem(cell_loss.sum().item(), w.sum().item())
        loss = cell_loss.sum() / w.sum()
        if clf.regularization.edge_epoch is not None:
            if clf.graph.additional_num_hops != 1:
                print("ERROR: clf.graph.additional_num_hops has to be >= 1 to use regularization")
                sys.exit(1)
            if clf.temp.current_epoch >= clf.regularization.edge_epoch:
                loss = loss + self.calcRegularization(logits_cell, data, clf, metrics)
        return loss

    def train(self, data_train, optimizer, clf):
        """One optimisation step on one sampled batch, reference :264-282."""
        self.model.train()
        logits_cell = self.model(data_train)
        n_sup = data_train.batch_adjs[self.model.num_layers - 1].size[1] if hasattr(data_train.batch_adjs[0], "size") \
            else data_train.batch_adjs[self.model.num_layers - 1][2][1]
        ids = data_train.batch_n_id[:n_sup].to(data_train.all.x.device)
        data_train.batch_x = data_train.all.x[ids]
        data_train.batch_gt = data_train.all.y[ids]
        loss = self.calcLossAndOA(logits_cell, None, data_train, clf, clf.training.metrics)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        return loss.detach()

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
