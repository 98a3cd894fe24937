"""Graph plans: the destination-sorted CSR (and its transpose) of a reference edge_index.

The reference hands every conv a dense ``LongTensor edge_index [2,E]`` (row 0 = source j, row 1 =
target i, local ids; processing/data.py:434-438, surfaceNetStaticEdgeFilters.py:217,262,303,344) and
lets PyG/torch_scatter scatter by ``edge_index[1]``.  A ``GraphPlan`` is built once per edge_index
on the GPU (dgnn_plan_build) and reused by all layers of a forward and by the backward.
"""
from __future__ import annotations

import weakref

import torch

from . import ops


class GraphPlan:
    """rowptr/src/eid: in-edges of destination i are sorted edges rowptr[i]..rowptr[i+1]-1, in
    ascending original edge position (the CPU scatter order of the reference); src[k] is the
    source of sorted edge k and eid[k] its row in the caller's edge_attr."""

    def __init__(self, edge_index: torch.Tensor, n_src: int, n_dst: int, hint: int = ops.PLAN_HINT_AUTO, parts=None):
        """`parts` = (rowptr int32 [n_dst+1], src int32 [E], eid int32 [E]) when the producer of the edge list already has
        them (the k-hop block builder emits its blocks grouped by destination: its own row offsets ARE the plan)."""
        if edge_index.dtype != torch.int64:
            edge_index = edge_index.to(torch.int64)  # inference_layer does the same (:339)
        self.edge_index = edge_index  # any strides: the plan builder reads the (possibly transposed) view in place
        self.n_src, self.n_dst, self.E = int(n_src), int(n_dst), int(edge_index.size(1))
        self.rowptr, self.src, self.eid = parts if parts is not None else ops.plan_build(self.edge_index, self.n_dst, by=1, hint=hint)
        self._grouped = parts is not None or hint == ops.PLAN_HINT_GROUPED
        self._t = None
        self._sorted_attr = None  # (weakref to edge_attr, version, sorted copy)

    @property
    def transposed(self):
        """(t_rowptr [n_src+1], t_dst [E], t_eid [E]): out-edges of every source, ascending edge
        position -- the accumulation order of autograd's index_add_ for x.index_select(0, src)."""
        if self._t is None:
            # an edge list grouped by destination is not grouped by source (unless it is the reference layout): skip the
            # fast-path attempts there
            self._t = ops.plan_build(self.edge_index, self.n_src, by=0, hint=ops.PLAN_HINT_GENERIC if self._grouped else ops.PLAN_HINT_AUTO)
        return self._t

    def sorted_edge_attr(self, edge_attr: torch.Tensor) -> torch.Tensor:
        """edge_attr rows permuted into plan order (one gather per scene, reused by every layer)."""
        c = self._sorted_attr
        if c is not None and c[0]() is edge_attr and c[1] == edge_attr._version:
            return c[2]
        out = ops.gather_rows(edge_attr, self.eid)
        self._sorted_attr = (weakref.ref(edge_attr), edge_attr._version, out)
        return out


_cache: dict = {}


def plan_for(edge_index: torch.Tensor, n_src: int, n_dst: int, cache: bool = True, hint: int = ops.PLAN_HINT_AUTO) -> GraphPlan:
    """Plan lookup keyed on the identity of the edge_index storage (training reuses the same block
    adjacency for forward and backward; inference reuses it across the 4 layers)."""
    if not cache:
        return GraphPlan(edge_index, n_src, n_dst, hint)
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, int(n_src), int(n_dst), str(edge_index.device))
    hit = _cache.get(key)
    if hit is not None and hit[0]() is edge_index:
        return hit[1]
    plan = GraphPlan(edge_index, n_src, n_dst, hint)
    if len(_cache) > 64:
        _cache.clear()
    try:
        _cache[key] = (weakref.ref(edge_index), plan)
    except TypeError:
        pass
    return plan


def register_plan(edge_index: torch.Tensor, plan: GraphPlan) -> None:
    """Makes plan_for(edge_index, n_src, n_dst) return `plan` (used by producers that build the plan with the edge list)."""
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, plan.n_src, plan.n_dst, str(edge_index.device))
    if len(_cache) > 64:
        _cache.clear()
    _cache[key] = (weakref.ref(edge_index), plan)


def clear_plan_cache():
    _cache.clear()
