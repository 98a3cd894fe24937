"""Oracle (test-only) restatement of learning/surfaceNetStaticEdgeFilters.py on plain PyTorch CPU.

Follows the reference line by line; every method cites the lines it restates.  State-dict keys
and shapes are those of the shipped checkpoint data/models/kf96/model_best.ptm.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import Linear

from .pyg_semantics import BatchNorm, propagate_mean


class SAGEConv(nn.Module):
    """learning/surfaceNetStaticEdgeFilters.py:20-109."""

    def __init__(self, lin_i, lin_j, lin_e):  # :46-51
        super().__init__()
        self.lin_i = lin_i
        self.lin_j = lin_j
        self.lin_e = lin_e

    def forward(self, x, edge_attr, edge_index, size=None):  # :66-87
        if isinstance(x, torch.Tensor):
            x = (x, x)  # :69-70
        phi = self.lin_e(edge_attr) if self.lin_e is not None else None  # :75-78
        out = propagate_mean(x[0], x[1].size(0), edge_index, phi)  # :80 (+ message :89-96)
        out = self.lin_j(out)  # :81
        x_r = x[1]
        if x_r is not None:
            out = out + self.lin_i(x_r)  # :84-86 (out += ...)
        return out


class SurfaceNet(nn.Module):
    """learning/surfaceNetStaticEdgeFilters.py:114-355."""

    def normLayer(self, size):  # :116-123
        if self.norm_type == 'b':
            return BatchNorm(size)
        elif self.norm_type == 'l':
            return nn.LayerNorm(size)  # PyG LayerNorm is graph-wise; never used by a shipped config
        return None

    def sageLayer(self, inp, out):  # :125-140
        li = Linear(inp, out, bias=False)
        lj = Linear(inp, out, bias=True)
        if self.clf.model.edge_convs == 1:
            le = Linear(self.n_edge_feat, inp, bias=True)
        elif self.clf.model.edge_convs == 2:
            le = nn.Sequential()
            le.add_module("0", Linear(self.n_edge_feat, int(self.n_edge_feat * 2)))
            le.add_module("1", self.normLayer(int(self.n_edge_feat * 2)))
            le.add_module("2", nn.ReLU(True))
            le.add_module("3", Linear(int(self.n_edge_feat * 2), inp))
        else:
            le = None
        return SAGEConv(li, lj, le)

    def __init__(self, clf):  # :146-187
        super().__init__()
        self.clf = clf
        self.n_classes = 2
        self.n_node_feat = clf.temp.num_node_features
        self.n_edge_feat = clf.temp.num_edge_features
        self.norm_type = clf.model.normalization
        self.output_dim = 2 if clf.training.loss == "kl" else 1  # :154-157
        self.convs = nn.ModuleList()
        widths = [self.n_node_feat] + list(clf.model.convs)
        for i in range(len(widths) - 1):  # :160-176
            blk = nn.Sequential()
            blk.add_module("conv", self.sageLayer(widths[i], widths[i + 1]))
            blk.add_module("norm", self.normLayer(widths[i + 1]))
            blk.add_module("relu", nn.ReLU(True))
            self.convs.append(blk)
        self.num_layers = len(self.convs)
        self.decoder = nn.Sequential()  # :180-187
        last = clf.model.convs[-1]
        if clf.model.decoder == 1:
            self.decoder.add_module("0", nn.Linear(last, self.output_dim))
        elif clf.model.decoder == 2:
            self.decoder.add_module("0", nn.Linear(last, int(last / 2)))
            self.decoder.add_module("1", self.normLayer(int(last / 2)))
            self.decoder.add_module("2", nn.ReLU(True))
            self.decoder.add_module("3", nn.Linear(int(last / 2), self.output_dim))

    # ---- train forward, :196-227 -------------------------------------------------------------
    def forward(self, data, trace=None):
        if self.clf.regularization.cell_type:
            x = data.all.x[data.batch_n_id, 1:]
        else:
            x = data.all.x[data.batch_n_id, :]
        for i in range(self.num_layers):
            edge_index, e_id, size = data.batch_adjs[i]
            x = self.convs[i][0]((x, x[:size[1]]), data.all.edge_attr[e_id], edge_index)
            if trace is not None:
                trace.append(("conv%d" % i, x.detach().clone()))
            x = self.convs[i][1](x)
            x = self.convs[i][2](x)
            if trace is not None:
                trace.append(("relu%d" % i, x.detach().clone()))
        if self.clf.model.decoder:
            x = self.decoder(x)
        return x

    # ---- inference, whole graph, :323-355 ----------------------------------------------------
    def inference_layer(self, data_all, trace=None):
        x = data_all.x[:, 1:] if self.clf.regularization.cell_type else data_all.x[:, :]
        xe = data_all.edge_attr[:, 1:] if self.clf.regularization.edge_type else data_all.edge_attr
        edge_index = data_all.edge_index.to(torch.long)
        for i in range(self.num_layers):
            x = self.convs[i][0]((x, x), xe, edge_index)
            if trace is not None:
                trace.append(("conv%d" % i, x.detach().clone()))
            x = self.convs[i][1](x)
            x = self.convs[i][2](x)
            if trace is not None:
                trace.append(("relu%d" % i, x.detach().clone()))
        if self.clf.model.decoder:
            x = self.decoder(x)
        return x

    # ---- inference, batch-major with k-hop blocks, :232-275 ----------------------------------
    def inference_batch_layer(self, data_all, batch_loader):
        x_out = torch.zeros([data_all.x.size(0), 2 if self.clf.training.loss == "kl" else 1], dtype=torch.float32)
        x_all = data_all.x[:, 1:] if self.clf.regularization.cell_type else data_all.x
        xe = data_all.edge_attr[:, 1:] if self.clf.regularization.edge_type else data_all.edge_attr
        for batch_size, n_id, adjs in batch_loader:
            x = x_all[n_id, :]
            for i in range(self.num_layers):
                edge_index, e_id, size = adjs[i]
                x = self.convs[i][0]((x, x[:size[1]]), xe[e_id], edge_index)
                x = self.convs[i][1](x)
                x = self.convs[i][2](x)
            if self.clf.model.decoder:
                x = self.decoder(x)
            x_out[n_id[:batch_size]] = x
        return x_out

    # ---- inference, layer-major with 1-hop blocks, :279-320 ----------------------------------
    def inference_layer_batch(self, data_all, batch_loader):
        x_all = data_all.x[:, 1:] if self.clf.regularization.cell_type else data_all.x
        xe = data_all.edge_attr[:, 1:] if self.clf.regularization.edge_type else data_all.edge_attr
        for i in range(self.num_layers):
            xs = []
            for batch_size, n_id, adj in batch_loader:
                edge_index, e_id, size = adj
                x = x_all[n_id]
                x = self.convs[i][0]((x, x[:size[1]]), xe[e_id], edge_index)
                x = self.convs[i][1](x)
                x = self.convs[i][2](x)
                xs.append(x)
            x_all = torch.cat(xs, dim=0)
        return self.decoder(x_all) if self.clf.model.decoder else x_all
