"""Restated semantics of the third-party calls on the dgnn hot path (oracle; test-only).

None of these packages is vendored in /root/reference (environment.yml:75,83-87 pins
pyg=2.0.2, pytorch-scatter=2.0.9, pytorch-sparse=0.6.12); what follows is the published
behaviour of those versions for exactly the call patterns the reference uses.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """torch_scatter 2.0.9 ``scatter(src, index, dim=0, dim_size=..., reduce='mean')``.

    out = zeros.scatter_add_(src); count = zeros.scatter_add_(ones); count[count<1] = 1;
    out /= count.  On CPU scatter_add_ accumulates in ascending position of ``src`` rows,
    which is the per-destination summation order the parity tests pin.
    Call site in the reference: the ``aggr='mean'`` of
    learning/surfaceNetStaticEdgeFilters.py:47 reached through ``self.propagate`` at :80.
    """
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out.scatter_add_(0, idx, src)
    count = torch.zeros(dim_size, dtype=src.dtype)
    count.scatter_add_(0, index, torch.ones(index.numel(), dtype=src.dtype))
    count[count < 1] = 1
    out.true_divide_(count.view(-1, *([1] * (src.dim() - 1))))
    return out


def propagate_mean(x_src: torch.Tensor, n_dst: int, edge_index: torch.Tensor, phi: torch.Tensor | None) -> torch.Tensor:
    """PyG 2.0.2 ``MessagePassing.propagate`` for a dense LongTensor ``edge_index``,
    flow source_to_target, node_dim=-2, aggr='mean', with the reference's ``message``
    (learning/surfaceNetStaticEdgeFilters.py:89-96): x_j = x_src[edge_index[0]];
    m = x_j * phi (or x_j); aggregate by edge_index[1] into ``n_dst`` rows; update = identity.
    """
    x_j = x_src.index_select(0, edge_index[0])
    m = x_j * phi if phi is not None else x_j
    return scatter_mean(m, edge_index[1], n_dst)


class BatchNorm(nn.Module):
    """``torch_geometric.nn.norm.BatchNorm``: wraps BatchNorm1d as ``.module`` (hence the
    checkpoint keys ``convs.N.norm.module.*``), eps 1e-5, momentum 0.1, affine, running stats."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True)

    def forward(self, x):
        return self.module(x)


def neighbor_sampler_full(edge_index: np.ndarray, n_nodes: int, batch: np.ndarray, num_hops: int):
    """PyG 2.0.2 ``NeighborSampler(edge_index, sizes=[-1]*num_hops, node_idx=batch, ...)`` for ONE
    batch, full neighbourhoods (size -1), ``return_e_id=True``  (call sites run.py:72-74, 221-223).

    adj_t is the CSR by destination with sources ascending (ties keep edge order);
    ``sample_adj(n_id, -1)`` returns, for the current target list ``n_id``, all in-edges in
    (target order, source ascending) order, and the new ``n_id`` = old ``n_id`` followed by newly
    seen sources in order of first appearance.  Blocks are emitted innermost first and the list
    is reversed, so ``adjs[0]`` is the outermost hop.  Returns
    ``(n_id, [(edge_index_local[2,E_l] int64, e_id[E_l] int64, (n_src, n_dst)), ...])``.
    """
    src, dst = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    E = src.shape[0]
    # CSR by destination, columns (sources) ascending, stable
    order = np.lexsort((np.arange(E), src, dst))
    rowptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    col = src[order]
    eid = order
    n_id = np.asarray(batch, dtype=np.int64)
    adjs = []
    for _ in range(num_hops):
        n_dst = n_id.shape[0]
        rows_l, cols_g, e_l = [], [], []
        for li, g in enumerate(n_id):
            s, e = rowptr[g], rowptr[g + 1]
            rows_l.append(np.full(e - s, li, dtype=np.int64))
            cols_g.append(col[s:e])
            e_l.append(eid[s:e])
        rows_l = np.concatenate(rows_l) if rows_l else np.zeros(0, np.int64)
        cols_g = np.concatenate(cols_g) if cols_g else np.zeros(0, np.int64)
        e_l = np.concatenate(e_l) if e_l else np.zeros(0, np.int64)
        # relabel sources: existing n_id keep their position, new ones appended in first-seen order
        pos = {int(g): i for i, g in enumerate(n_id)}
        new = []
        cols_l = np.empty_like(cols_g)
        for k, g in enumerate(cols_g):
            g = int(g)
            p = pos.get(g)
            if p is None:
                p = len(pos)
                pos[g] = p
                new.append(g)
            cols_l[k] = p
        n_id = np.concatenate([n_id, np.asarray(new, dtype=np.int64)])
        adjs.append((np.stack([cols_l, rows_l]), e_l, (n_id.shape[0], n_dst)))
    return n_id, adjs[::-1]
