"""GPU k-hop full-neighbour block builder: drop-in for the CPU
``torch_geometric.data.NeighborSampler(edge_index, sizes=[-1]*k, node_idx, batch_size, shuffle, drop_last,
return_e_id=True)`` that the reference builds at run.py:72-74 (training) and run.py:221-223 (inference).

Iterating yields ``(batch_size, n_id, adjs)`` exactly as PyG does: ``adjs`` is the list of
``(edge_index [2,E_l] local ids, e_id [E_l], size=(n_src, n_dst))`` with the OUTERMOST hop first (a single
triple when there is one hop, as consumed at surfaceNetStaticEdgeFilters.py:300-301); targets are always a prefix
of sources.  Everything stays on the GPU (the reference samples on the CPU with torch_sparse and copies each block
over PCIe).  Per destination, edges come in plan order = ascending edge position; for the reference's adjacency
layout (4 rows per source tet, distinct neighbours) that is ascending source id, i.e. PyG's order, so blocks are
identical to the CPU sampler's (tests/test_gpu_parity.py checks this against the oracle's restatement).
"""
from __future__ import annotations

import torch

from ._lib import check, lib, ptr, stream_ptr
from .graph import GraphPlan, register_plan

_I32_MAX = 2 ** 31 - 1


class EdgeIndex(tuple):
    """(edge_index, e_id, size) with attribute access like PyG's namedtuple (calcRegularization reads
    ``batch_adjs[k].edge_index`` / ``.size``, learning/runModel.py:118-119)."""

    def __new__(cls, edge_index, e_id, size):
        return super().__new__(cls, (edge_index, e_id, size))

    edge_index = property(lambda self: self[0])
    e_id = property(lambda self: self[1])
    size = property(lambda self: self[2])

    def to(self, *a, **k):
        return EdgeIndex(self[0].to(*a, **k), self[1].to(*a, **k) if self[1] is not None else None, self[2])


class NeighborSampler:
    def __init__(self, edge_index, sizes, node_idx=None, num_nodes=None, batch_size=1, shuffle=False, drop_last=False,
                 return_e_id=True, plan: GraphPlan = None, generator=None, prefetch=True, **kwargs):
        if any(int(s) != -1 for s in sizes):
            raise NotImplementedError("only full neighbourhoods (size -1) are used by the reference (clique_sizes: [-1])")
        if not edge_index.is_cuda:
            raise RuntimeError("dgnn_amd.sampler.NeighborSampler builds blocks on the GPU: pass a CUDA edge_index")
        self.sizes = list(sizes)
        self.device = edge_index.device
        n = int(num_nodes) if num_nodes is not None else int(edge_index.max().item()) + 1
        self.num_nodes = n
        self.plan = plan if plan is not None else GraphPlan(edge_index, n, n)
        if node_idx is None:
            node_idx = torch.arange(n, device=self.device)
        elif node_idx.dtype == torch.bool:
            node_idx = node_idx.nonzero(as_tuple=False).view(-1)
        self.node_idx = node_idx.to(self.device, torch.int64)
        self.batch_size, self.shuffle, self.drop_last, self.return_e_id = int(batch_size), shuffle, drop_last, return_e_id
        self.generator = generator
        # iteration builds block k+1 on a side stream while the caller's stream still runs step k-1 / k (the reference's
        # NeighborSampler is a DataLoader: its workers prefetch batches the same way); the builder's host round trips then
        # wait for the side stream only, not for the training step's kernels
        self.prefetch = bool(prefetch)
        self._side = None
        self._escaped = None
        self._pos = torch.full((n,), -1, dtype=torch.int32, device=self.device)
        self._first = torch.full((n,), _I32_MAX, dtype=torch.int32, device=self.device)
        self._iota = torch.arange(0, dtype=torch.int32, device=self.device)
        # every cell of a Delaunay scene has exactly 4 in-edges: a block's edge count is then 4 x targets and one of the
        # two host round trips per hop (reading the count back) is not needed
        rp = self.plan.rowptr
        self._regular = bool(((rp[1:] - rp[:-1]) == 4).all().item()) if n > 0 else False

    def _arange(self, n):
        if self._iota.numel() < n:
            self._iota = torch.arange(max(n, 2 * self._iota.numel()), dtype=torch.int32, device=self.device)
        return self._iota[:n]

    def __len__(self):
        m = self.node_idx.numel()
        return m // self.batch_size if self.drop_last else (m + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        idx = self.node_idx
        if self.shuffle:
            idx = idx[torch.randperm(idx.numel(), device=self.device, generator=self.generator)]
        starts = [s for s in range(0, idx.numel(), self.batch_size)
                  if not (self.drop_last and idx.numel() - s < self.batch_size)]
        if not self.prefetch:
            for s in starts:
                yield self.sample(idx[s:s + self.batch_size])
            return
        with torch.cuda.device(self.device):
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
            side = self._side
            side.wait_stream(torch.cuda.current_stream())   # idx (and the builder state) were produced on the caller's stream
        pending = None
        for s in starts:
            with torch.cuda.device(self.device), torch.cuda.stream(side):
                self._escaped = []
                out = self._sample(idx[s:s + self.batch_size])
                nxt = (out, self._escaped, side.record_event())
                self._escaped = None
            if pending is not None:
                yield self._hand_over(pending)
            pending = nxt
        if pending is not None:
            yield self._hand_over(pending)

    def _hand_over(self, pending):
        """makes a block built on the side stream usable on the caller's current stream"""
        out, tensors, ev = pending
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for t in tensors:
                t.record_stream(cur)   # the allocator must not recycle them for the next block while this stream reads them
        return out

    def sample(self, batch: torch.Tensor):
        with torch.cuda.device(self.device):  # the plan's GPU, whatever the thread's current device is
            return self._sample(batch)

    def _sample(self, batch: torch.Tensor):
        L, st, p = lib(), stream_ptr(), self.plan
        n_id = batch.to(self.device, torch.int64).contiguous()
        batch_size = n_id.numel()
        adjs = []
        for hop in range(len(self.sizes)):
            n_t = n_id.numel()
            off = torch.empty(n_t + 1, dtype=torch.int32, device=self.device)
            scratch = torch.empty(int(L.dgnn_khop_scratch_elems(n_t, 0)), dtype=torch.int32, device=self.device)
            check(L.dgnn_khop_count(ptr(p.rowptr), ptr(n_id), n_t, int(hop == 0), ptr(self._pos), ptr(off), ptr(scratch), st),
                  "dgnn_khop_count")
            n_e = 4 * n_t if self._regular else int(off[n_t].item())  # sizes the block tensors
            ei = torch.empty((2, n_e), dtype=torch.int64, device=self.device)
            e_src, e_dst = ei[0], ei[1]
            e_id = torch.empty(n_e, dtype=torch.int64, device=self.device)
            n_id_out = torch.empty(n_t + n_e, dtype=torch.int64, device=self.device)
            n_new = torch.zeros(1, dtype=torch.int32, device=self.device)
            scratch = torch.empty(int(L.dgnn_khop_scratch_elems(n_t, n_e)), dtype=torch.int32, device=self.device)
            check(L.dgnn_khop_expand(ptr(p.rowptr), ptr(p.src), ptr(p.eid), ptr(n_id), n_t, ptr(off), n_e, ptr(self._pos),
                                     ptr(self._first), ptr(e_src), ptr(e_dst), ptr(e_id), ptr(n_id_out), ptr(n_new), ptr(scratch),
                                     st), "dgnn_khop_expand")
            n_all = n_t + int(n_new.item())
            check(L.dgnn_khop_commit(ptr(n_id_out), n_t, n_all, ptr(self._pos), ptr(self._first), st), "dgnn_khop_commit")
            n_id = n_id_out[:n_all]
            # the block is emitted grouped by destination with `off` as its row offsets: that IS its plan (identity order)
            src32 = e_src.to(torch.int32)
            register_plan(ei, GraphPlan(ei, n_all, n_t, parts=(off, src32, self._arange(n_e))))
            if self._escaped is not None:
                self._escaped += [ei, e_id, off, src32, self._iota, n_id_out]
            adjs.append(EdgeIndex(ei, e_id if self.return_e_id else None, (n_all, n_t)))
        check(L.dgnn_khop_reset(ptr(n_id), n_id.numel(), ptr(self._pos), st), "dgnn_khop_reset")
        adjs = adjs[0] if len(adjs) == 1 else adjs[::-1]
        return batch_size, n_id, adjs
