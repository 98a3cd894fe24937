"""Oracle (test-only) restatement of learning/surfaceNetUpdatedEdgeFilters.py on plain PyTorch CPU.

Only the coherent part of that (orphaned) file is restated: SAGEConv (:23-185) and
SurfaceNet.__init__/forward (:191-251).  Its three inference_* methods call the conv without
``edge_attr`` (:281,:317,:352) and raise in the reference; they are not part of the path.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Linear

from .pyg_semantics import propagate_mean


class SAGEConv(nn.Module):
    """learning/surfaceNetUpdatedEdgeFilters.py:23-185."""

    def __init__(self, in_channels, out_channels, edge_in_channels, normalize=False, bias=True):  # :45-63
        super().__init__()
        self.in_channels = in_channels
        self.edge_in_channels = edge_in_channels
        self.out_channels = out_channels
        self.normalize = normalize
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.lin_l = Linear(in_channels[0], out_channels, bias=bias)
        self.lin_r = Linear(in_channels[1], out_channels, bias=False)
        self.lin_e = Linear(edge_in_channels, in_channels[0], bias=bias)

    def forward(self, x, edge_attr, edge_index, size=None):  # :147-170
        if isinstance(x, torch.Tensor):
            x = (x, x)
        edge_attr = self.lin_e(edge_attr)  # :156
        out = propagate_mean(x[0], x[1].size(0), edge_index, edge_attr)  # :158 (+ :70-140, :172-176)
        out = self.lin_l(out)  # :159
        x_r = x[1]
        if x_r is not None:
            out = out + self.lin_r(x_r)  # :162-165
        if self.normalize:
            out = F.normalize(out, p=2., dim=-1)  # :167-168
        return out, edge_attr


class SurfaceNet(nn.Module):
    """learning/surfaceNetUpdatedEdgeFilters.py:188-251."""

    def __init__(self, n_node_features, clf):  # :191-210
        super().__init__()
        self.clf = clf
        self.n_classes = 2
        self.n_node_feat = n_node_features
        p = clf.training.model_params
        self.convs = nn.ModuleList()
        self.convs.append(SAGEConv(self.n_node_feat, p[0], 2))
        self.convs.append(SAGEConv(p[0], p[1], self.n_node_feat))
        for i in range(len(p) - 2):
            self.convs.append(SAGEConv(p[i + 1], p[i + 2], p[i], normalize=False))
        self.num_layers = len(self.convs)
        if clf.training.model_name[-1] == "+":
            self.out_net = nn.Sequential(nn.ReLU(True), nn.Linear(p[-1], 128), nn.ReLU(True), nn.Linear(128, 2))

    def forward(self, data_all, trace=None):  # :216-251
        f = self.clf.features
        if f.normalization_feature and not f.keep_normalization_feature:
            x = data_all.x[data_all.n_id, 1:]
        else:
            x = data_all.x[data_all.n_id, :]
        edge_attr = data_all.edge_attr
        for i in range(self.num_layers):
            edge_index, e_id, size = data_all.adjs[i]
            new_edge_attr = torch.zeros([data_all.edge_attr.shape[0], self.convs[i].in_channels])  # :236
            x, new_edge_attr[e_id] = self.convs[i]((x, x[:size[1]]), edge_attr[e_id, :self.convs[i].edge_in_channels], edge_index)  # :237
            edge_attr = new_edge_attr
            if trace is not None:
                trace.append(("conv%d" % i, x.detach().clone()))
                trace.append(("phi%d" % i, edge_attr.detach().clone()))
            if i != self.num_layers - 1:  # :239-241
                x = F.relu(x)
                edge_attr = F.relu(edge_attr)
        if self.clf.training.model_name[-1] == "+":  # :245-247
            x = F.relu(x)
            x = self.out_net(x)
        return x
