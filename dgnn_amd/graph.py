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
        converted = edge_index.dtype != torch.int64
        if converted:
            edge_index = edge_index.to(torch.int64)  # inference_layer does the same (:339)
        # any strides: the plan builder reads the (possibly transposed) view in place.  Held WEAKLY: the transposed plan is
        # built lazily from it (backward; the autograd nodes keep the tensor alive until then) and the plan itself hangs off
        # the tensor (plan_for), so a strong reference would be a cycle (a converted int64 copy has no other owner and is kept).
        self._ei_ref = weakref.ref(edge_index)
        self._ei_own = edge_index if converted else None
        self.n_src, self.n_dst, self.E = int(n_src), int(n_dst), int(edge_index.size(1))
        # Every array of a plan is held as a tensor OR as (buffer, length): the block builder hands out prefixes of preallocated buffers, and
        # cutting the ~12 views of a block costs the training loop's main thread more than the library calls that only need the ADDRESSES
        # (`*_ptrs`).  A view is cut when somebody asks for the tensor.
        self._parts = list(parts if parts is not None else ops.plan_build(edge_index, self.n_dst, by=1, hint=hint, n_other=self.n_src))
        self._grouped = parts is not None or hint == ops.PLAN_HINT_GROUPED
        self._t = None            # transposed plan: [t_rowptr, t_dst, t_eid], same convention
        self._sorted_attr = None  # (weakref to edge_attr, version, sorted copy)
        # optional, set by producers whose eid is the identity (the k-hop block builder): int32 [E], row of the SCENE's edge_attr
        # behind block edge k, so that kernels gather attribute rows from the scene tensor instead of a per-block copy
        self._edge_rows = None
        self._t_rows = None

    @staticmethod
    def _tensor(holder, i):
        v = holder[i]
        if isinstance(v, tuple):
            v = holder[i] = v[0][:v[1]]
        return v

    @staticmethod
    def _address(v):
        return (v[0] if isinstance(v, tuple) else v).data_ptr()      # a prefix starts where its buffer starts

    rowptr = property(lambda self: self._tensor(self._parts, 0))
    src = property(lambda self: self._tensor(self._parts, 1))
    eid = property(lambda self: self._tensor(self._parts, 2))

    @property
    def edge_rows(self):
        if isinstance(self._edge_rows, tuple):
            self._edge_rows = self._edge_rows[0][:self._edge_rows[1]]
        return self._edge_rows

    @edge_rows.setter
    def edge_rows(self, v):
        self._edge_rows = v

    @property
    def has_edge_rows(self):
        return self._edge_rows is not None

    def part_ptrs(self, scene_rows: bool = False):
        """addresses of (rowptr, src, eid | edge_rows) for library calls; no views are cut"""
        return (self._address(self._parts[0]), self._address(self._parts[1]), self._address(self._edge_rows if scene_rows else self._parts[2]))

    def transposed_ptrs(self, scene_rows: bool = False):
        """addresses of (t_rowptr, t_dst, t_eid | transposed edge_rows); builds the transposed plan when the producer did not supply it"""
        if self._t is None or (scene_rows and self._t_rows is None):
            t = self.transposed
            return (t[0].data_ptr(), t[1].data_ptr(), (self.transposed_edge_rows if scene_rows else t[2]).data_ptr())
        return (self._address(self._t[0]), self._address(self._t[1]), self._address(self._t_rows if scene_rows else self._t[2]))

    @property
    def edge_index(self):
        """The edge list this plan was built from (None once its owner has freed it)."""
        return self._ei_own if self._ei_own is not None else self._ei_ref()

    @property
    def transposed(self):
        """(t_rowptr [n_src+1], t_dst [E], t_eid [E]): out-edges of every source, ascending edge
        position -- the accumulation order of autograd's index_add_ for x.index_select(0, src)."""
        if self._t is None:
            edge_index = self.edge_index
            if edge_index is None:
                raise RuntimeError("GraphPlan.transposed: the edge_index tensor this plan was built from has been freed")
            # an edge list grouped by destination is not grouped by source (unless it is the reference layout): skip the
            # fast-path attempts there
            self._t = list(ops.plan_build(edge_index, self.n_src, by=0, hint=ops.PLAN_HINT_GENERIC if self._grouped else ops.PLAN_HINT_AUTO,
                                           n_other=self.n_dst))
        return tuple(self._tensor(self._t, i) for i in range(3))

    @property
    def transposed_edge_rows(self):
        """edge_rows in the transposed plan's order (None without edge_rows)"""
        if self._t_rows is None and self.edge_rows is not None:
            self._t_rows = torch.index_select(self.edge_rows, 0, self.transposed[2])
        if isinstance(self._t_rows, tuple):
            self._t_rows = self._t_rows[0][:self._t_rows[1]]
        return self._t_rows

    def sorted_edge_attr(self, edge_attr: torch.Tensor) -> torch.Tensor:
        """edge_attr rows permuted into plan order (one gather per scene, reused by every layer)."""
        c = self._sorted_attr
        if c is not None and c[0]() is edge_attr and c[1] == edge_attr._version:
            return c[2]
        out = ops.gather_rows(edge_attr, self.eid)
        self._sorted_attr = (weakref.ref(edge_attr), edge_attr._version, out)
        return out


def _key(edge_index: torch.Tensor, n_src: int, n_dst: int):
    return (int(n_src), int(n_dst), edge_index._version, edge_index.data_ptr(), tuple(edge_index.shape), tuple(edge_index.stride()))


def plan_for(edge_index: torch.Tensor, n_src: int, n_dst: int, cache: bool = True, hint: int = ops.PLAN_HINT_AUTO) -> GraphPlan:
    """Plan lookup.  The plan lives ON the caller's edge_index tensor object (training reuses the same block adjacency for
    forward and backward; inference reuses it across the 4 layers; a resident scene across calls) and dies with it: there
    is no global table that could pin rowptr/src/eid -- or a plan-ordered edge_attr copy -- in HBM after the graph is gone.
    A tensor that arrives on the CPU is copied to the GPU by the model on every call; its plan then lasts one call."""
    if not cache:
        return GraphPlan(edge_index, n_src, n_dst, hint)
    key = _key(edge_index, n_src, n_dst)
    held = getattr(edge_index, "_dgnn_plans", None)
    if held is not None and key in held:
        return held[key]
    plan = GraphPlan(edge_index, n_src, n_dst, hint)
    register_plan(edge_index, plan)
    return plan


def register_plan(edge_index: torch.Tensor, plan: GraphPlan) -> None:
    """Makes plan_for(edge_index, n_src, n_dst) return `plan` (used by producers that build the plan with the edge list)."""
    key = _key(edge_index, plan.n_src, plan.n_dst)
    held = getattr(edge_index, "_dgnn_plans", None)
    if held is None or any(k[2] != key[2] for k in held):   # first plan, or the tensor was written to since
        held = {}
    held[key] = plan
    try:
        edge_index._dgnn_plans = held
    except AttributeError:
        pass


def clear_plan_cache(edge_index: torch.Tensor = None):
    """Drops the plans attached to `edge_index` (nothing global exists any more; kept for callers of the old API)."""
    if edge_index is not None and hasattr(edge_index, "_dgnn_plans"):
        del edge_index._dgnn_plans
