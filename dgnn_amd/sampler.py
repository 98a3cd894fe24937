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
ONE_CALL = __import__("os").environ.get("DGNN_KHOP_ONE_CALL", "1") != "0"   # regular graphs: all hops in one library call


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
                 return_e_id=True, plan: GraphPlan = None, generator=None, prefetch=True, transposed_plans=True, reuse_buffers=False,
                 **kwargs):
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
        self.prefetch = "thread" if prefetch == "thread" else bool(prefetch)   # False | True (side stream, this thread) | "thread"
        self._side = None
        # prefetching iteration also builds every block's source-sorted (transposed) plan, which the backward pass needs, on the
        # builder's stream; False leaves it to the first backward (inference loaders never need it)
        self.transposed_plans = bool(transposed_plans)
        # reuse_buffers (prefetching iteration over a 4-regular graph only): blocks are built into a ring of three preallocated buffer
        # sets instead of ~50 fresh tensors each -- a block (its n_id, edge lists, plans) is then valid only until TWO more blocks have
        # been drawn.  A training loop consumes a block before it asks for the next; anything that keeps blocks (list(loader)) must not
        # set it.
        self._ring = [None] * 3 if reuse_buffers else None
        self._rows = []
        self._escaped = None
        self._pos = torch.full((n,), -1, dtype=torch.int32, device=self.device)
        self._first = torch.full((n,), _I32_MAX, dtype=torch.int32, device=self.device)
        self._iota = torch.arange(0, dtype=torch.int32, device=self.device)
        # every cell of a Delaunay scene has exactly 4 in-edges: a block's edge count is then 4 x targets and one of the
        # two host round trips per hop (reading the count back) is not needed
        rp = self.plan.rowptr
        self._regular = bool(((rp[1:] - rp[:-1]) == 4).all().item()) if n > 0 else False

    def attach_rows(self, specs):
        """Rows the builder gathers behind every block, on its own stream (prefetching iteration with reuse_buffers over a 4-regular graph; other
        modes ignore it): `specs` = up to 4 tuples (src, col0, cols, which) -- src a resident fp32 [N, C] tensor with unit column stride, the columns
        [col0, col0 + cols) of its rows at the block's node ids (which = "all": n_id) or at the batch's targets (which = "batch": n_id[:batch_size]).
        What the reference's training loop indexes at the head of every step (x_all[n_id, 1:], x_all[ids], y_all[ids]; learning/
        surfaceNetStaticEdgeFilters.py:206, learning/runModel.py:273-274).  Consumers ask `block_rows(n_id, src, col0, cols, which)`."""
        ok = []
        for src, col0, cols, which in specs:
            if isinstance(src, torch.Tensor) and src.is_cuda and src.device == self.device and src.dtype == torch.float32 and src.dim() == 2 \
                    and src.stride(1) == 1 and 0 <= col0 and col0 + cols <= src.size(1) and cols > 0 and which in ("all", "batch"):
                ok.append((src, int(col0), int(cols), which))
        self._rows = ok[:4]
        if self._ring is not None:
            self._ring = [None] * 3        # buffer sets are sized with their row outputs (call this BEFORE iterating, not from inside the loop)

    def _arange_full(self, n):
        """the shared 0, 1, 2, ... buffer, at least n long"""
        if self._iota.numel() < n:
            self._iota = torch.arange(max(n, 2 * self._iota.numel()), dtype=torch.int32, device=self.device)
        return self._iota

    def _arange(self, n):
        return self._arange_full(n)[:n]

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
        if self.prefetch != "thread" and self._regular and ONE_CALL and self.batch_size > 0:
            # the library's own host thread builds block k+1 while this thread enqueues step k
            pending = None
            ring, ring_free, k = self._ring, [None] * 3, 0   # reuse_buffers: three buffer sets, block j builds into set j % 3
            with torch.cuda.device(self.device):
                self._caller_stream = torch.cuda.current_stream()
            # the loop below runs once per training step on the thread that issues the step: with the buffer ring nothing in it launches or
            # allocates, so it needs no stream / device context managers (the builder gets the side stream's handle) and its events are reused
            free_ev = [torch.cuda.Event() for _ in range(3)]
            self._done_ev, self._done_k = [torch.cuda.Event() for _ in range(4)], 0
            try:
                for s in starts:
                    blk = self._finish_on_side(pending) if pending is not None else None
                    pending = None
                    batch = idx[s:s + self.batch_size]      # (a row range of a contiguous vector: contiguous)
                    if not batch.is_contiguous():
                        with torch.cuda.device(self.device), torch.cuda.stream(side):
                            batch = batch.contiguous()
                    slot = None
                    if ring is not None:
                        # everything the caller enqueued up to now (the step on block k-2 and before) precedes this point on its
                        # stream; set k % 3 was last handed out as block k-3: the builder may overwrite it once that point is passed
                        slot = k % 3
                        if ring[slot] is None or ring[slot]["cap_nb"] < batch.numel():
                            with torch.cuda.device(self.device), torch.cuda.stream(side):
                                ring[slot] = self._alloc_regular(max(self.batch_size, batch.numel()), self.transposed_plans)
                            for v in ring[slot].values():   # allocated on the builder's stream, read on the caller's for their whole life
                                for t in (v if isinstance(v, list) else [v]):
                                    if isinstance(t, torch.Tensor):
                                        t.record_stream(self._caller_stream)
                        if ring_free[slot] is not None:
                            side.wait_event(ring_free[slot])
                        self._escaped = None
                        # the library reads hipGetDevice() for its builder thread and the side stream belongs to self.device: make that device
                        # current around the call when the caller's thread sits on another one (a bare setDevice pair, no context-manager objects)
                        cur_dev, want_dev = torch._C._cuda_getDevice(), self.device.index
                        if cur_dev != want_dev:
                            torch._C._cuda_setDevice(want_dev)
                        try:
                            pending = (self._start_regular(batch, background=True, b=ring[slot], stream=side.cuda_stream), None)
                        finally:
                            if cur_dev != want_dev:
                                torch._C._cuda_setDevice(cur_dev)
                    else:
                        with torch.cuda.device(self.device), torch.cuda.stream(side):
                            self._escaped = []
                            pending = (self._start_regular(batch.contiguous(), background=True, b=None), self._escaped)
                            self._escaped = None
                    k += 1
                    if blk is not None:
                        if ring is not None:
                            # the caller is done enqueueing the step on the block before this one: mark that point on ITS stream
                            ev = free_ev[(k - 3) % 3]
                            ev.record(torch.cuda.current_stream(self.device))   # (the sampler's device, whatever the thread's current one is)
                            ring_free[(k - 3) % 3] = ev
                        yield self._hand_over(blk)
                if pending is not None:
                    blk, pending = self._finish_on_side(pending), None
                    yield self._hand_over(blk)
            finally:
                if pending is not None:   # abandoned mid-way (also at interpreter exit): the builder thread must not outlive its buffers
                    b = pending[0]
                    job, b["job"] = b.get("job"), None
                    if job:
                        try:
                            lib().dgnn_khop_blocks_regular_wait(job, b["hops"], b["counts"])   # joins; no torch calls on this path
                        except Exception:  # noqa: BLE001 -- module globals may be gone during teardown
                            pass
            return
        if self.prefetch != "thread":
            pending = None
            for s in starts:
                nxt = self._build_on_side(idx[s:s + self.batch_size])
                if pending is not None:
                    yield self._hand_over(pending)
                pending = nxt
            if pending is not None:
                yield self._hand_over(pending)
            return
        # a worker thread builds blocks ahead (queue of 2): its host round trips (one per hop) wait while this thread keeps
        # enqueueing the training step -- ctypes calls and .item() release the GIL
        import queue
        import threading
        q, stop = queue.Queue(maxsize=2), threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def worker():
            try:
                for s in starts:
                    if stop.is_set() or not put(self._build_on_side(idx[s:s + self.batch_size])):
                        return
                put(None)
            except BaseException as e:   # surfaces in the consuming thread
                put(e)

        t = threading.Thread(target=worker, name="dgnn-block-builder", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield self._hand_over(item)
        finally:
            stop.set()
            t.join()

    def _finish_on_side(self, pending):
        b, escaped = pending
        if escaped is None and getattr(self, "_done_ev", None) is not None:
            # buffer ring: joining the builder and cutting the views launches nothing -- no stream switch; the event marks the side stream's tail
            out = self._finish_regular(b)
            ev = self._done_ev[self._done_k & 3]
            self._done_k += 1
            ev.record(self._side)
            return (out, None, ev)
        with torch.cuda.device(self.device), torch.cuda.stream(self._side):
            self._escaped = escaped
            try:
                out = self._finish_regular(b)
            finally:
                self._escaped = None
            return (out, escaped, self._side.record_event())

    def _build_on_side(self, batch):
        with torch.cuda.device(self.device), torch.cuda.stream(self._side):
            self._escaped = []
            out = self._sample(batch)
            item = (out, self._escaped, self._side.record_event())
            self._escaped = None
        return item

    def _hand_over(self, pending):
        """makes a block built on the side stream usable on the caller's current stream"""
        out, tensors, ev = pending
        if not tensors:                  # the block lives in the reuse ring (or escaped nothing): only the ordering is needed
            torch.cuda.current_stream(self.device).wait_event(ev)
            return out
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for t in tensors:
                t.record_stream(cur)   # the allocator must not recycle them for the next block while this stream reads them
        return out

    def sample(self, batch: torch.Tensor):
        with torch.cuda.device(self.device):  # the plan's GPU, whatever the thread's current device is
            return self._sample(batch)

    def _alloc_regular(self, nb: int, want_t: bool):
        """buffers of one batch of `nb` targets, sized by the 5^h growth bound (4-regular graph: 4 new sources per target at most),
        and the argument arrays that only depend on them"""
        import ctypes as C
        hops, dev = len(self.sizes), self.device
        cap_t = [min(nb * 5 ** h, self.num_nodes) for h in range(hops)]
        cap_e = [4 * t for t in cap_t]
        i64 = lambda n: torch.empty(n, dtype=torch.int64, device=dev)      # nb > 0: every capacity is positive
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
        b = dict(cap_nb=nb, hops=hops, want_t=want_t,
                 ei=[torch.empty((2, e), dtype=torch.int64, device=dev) for e in cap_e], e_id=[i64(e) for e in cap_e],
                 src32=[i32(e) for e in cap_e], e_id32=[i32(e) for e in cap_e], off=[i32(t + 1) for t in cap_t],
                 n_out=[i64(t + e) for t, e in zip(cap_t, cap_e)])
        L = lib()
        b["scratch"] = scratch = i32(max(int(L.dgnn_khop_scratch_elems(t, e)) for t, e in zip(cap_t, cap_e)) + 1)
        arr = lambda ts: (C.c_void_p * hops)(*[t.data_ptr() for t in ts])
        caps = lambda v: (C.c_int64 * hops)(*v)
        t_arrs, cap_all, plan_scratch = [None] * 4, None, None
        if want_t:   # prefetching iteration (training): the source-sorted plans the backward pass walks come out of the same call
            cap_all = [min(t + e, self.num_nodes) for t, e in zip(cap_t, cap_e)]
            b["t_rowptr"] = [i32(a + 1) for a in cap_all]
            b["t_dst"], b["t_eid"], b["t_rows"] = ([i32(e) for e in cap_e] for _ in range(3))
            b["plan_scratch"] = plan_scratch = i32(max(int(L.dgnn_plan_scratch_elems(e, a)) for e, a in zip(cap_e, cap_all)))
            t_arrs = [arr(b[k]) for k in ("t_rowptr", "t_dst", "t_eid", "t_rows")]
        b["args_tail"] = (arr(b["ei"]), arr(b["e_id"]), arr(b["src32"]), arr(b["e_id32"]), arr(b["off"]), arr(b["n_out"]), caps(cap_t), caps(cap_e),
                          ptr(scratch[:-1]), ptr(scratch[-1:]), *t_arrs, caps(cap_all) if want_t else None, ptr(plan_scratch))
        if self._rows:     # row gathers behind the block (attach_rows): outputs sized by the outermost block's bound / the batch
            cap_nodes = min(cap_t[-1] + cap_e[-1], self.num_nodes)
            outs = [torch.empty((cap_nodes if which == "all" else nb, cols), dtype=torch.float32, device=dev) for _, _, cols, which in self._rows]
            k = len(outs)
            b["rows_out"] = outs
            b["rows_spec"] = list(self._rows)
            b["rows_args"] = (k, (C.c_void_p * k)(*[src.data_ptr() + 4 * col0 for src, col0, _, _ in self._rows]),
                              (C.c_int64 * k)(*[src.stride(0) for src, _, _, _ in self._rows]), (C.c_int32 * k)(*[cols for _, _, cols, _ in self._rows]),
                              (C.c_int32 * k)(*[int(which == "batch") for _, _, _, which in self._rows]), (C.c_void_p * k)(*[o.data_ptr() for o in outs]))
        return b

    def _start_regular(self, n_id: torch.Tensor, background: bool, b=None, stream=None):
        """every node has exactly 4 in-edges: all hops (and, when iterating with prefetch, the transposed plans) in one library
        call.  `background`: the call runs on a library-owned host thread (dgnn_khop_blocks_regular_start) and this returns at
        once; _finish_regular joins it and cuts the views.  `b`: buffers to build into (a slot of the reuse ring), default fresh ones."""
        import ctypes as C
        p, nb = self.plan, n_id.numel()
        if b is None:
            b = self._alloc_regular(nb, self.transposed_plans and self._escaped is not None)
        b["nb"], b["n_id"] = nb, n_id
        b["counts"] = counts = (C.c_int64 * (b["hops"] + 1))()
        L = lib()
        args = (ptr(p.rowptr), ptr(p.src), ptr(p.eid), 4, ptr(n_id), nb, b["hops"], ptr(self._pos), ptr(self._first)) + b["args_tail"]
        if background and b.get("rows_args") is not None:
            b["job"] = L.dgnn_khop_blocks_regular_start_rows(*args, *b["rows_args"], stream if stream is not None else stream_ptr())
            if not b["job"]:
                check(-1, "dgnn_khop_blocks_regular_start_rows")
            b["rows_live"] = True
        elif background:
            b["rows_live"] = False
            b["job"] = L.dgnn_khop_blocks_regular_start(*args, stream if stream is not None else stream_ptr())
            if not b["job"]:
                check(-1, "dgnn_khop_blocks_regular_start")
        else:
            b["rows_live"] = False
            check(L.dgnn_khop_blocks_regular(*args, counts, stream_ptr()), "dgnn_khop_blocks_regular", poll=True)
        return b

    def _finish_regular(self, b):
        hops, counts = b["hops"], b["counts"]
        if b.get("job"):
            job, b["job"] = b["job"], None
            check(lib().dgnn_khop_blocks_regular_wait(job, hops, counts), "dgnn_khop_blocks_regular", poll=True)
        adjs = []
        for h in range(hops):
            n_t, n_all = int(counts[h]), int(counts[h + 1])
            n_e = 4 * n_t
            e = b["ei"][h][:, :n_e]     # a strided view: plans and kernels read edge lists in place
            # (buffer, length) pairs: the plan cuts a view when a tensor is asked for; the training step's library calls take the addresses
            plan = GraphPlan(e, n_all, n_t, parts=((b["off"][h], n_t + 1), (b["src32"][h], n_e), (self._arange_full(n_e), n_e)))
            # plan position k <-> block edge k <-> row e_id[k] of the scene's edge_attr: consumers may read those rows in place
            plan.edge_rows = (b["e_id32"][h], n_e)
            if b["want_t"]:
                plan._t = [(b["t_rowptr"][h], n_all + 1), (b["t_dst"][h], n_e), (b["t_eid"][h], n_e)]
                plan._t_rows = (b["t_rows"][h], n_e)
                if self._escaped is not None:
                    self._escaped += [b[k][h] for k in ("t_rowptr", "t_dst", "t_eid", "t_rows")]
            register_plan(e, plan)
            adjs.append(EdgeIndex(e, b["e_id"][h][:n_e] if self.return_e_id else None, (n_all, n_t)))
            if self._escaped is not None:
                self._escaped += [b[k][h] for k in ("ei", "e_id", "off", "src32", "e_id32", "n_out")] + [self._iota]
        adjs = adjs[0] if len(adjs) == 1 else adjs[::-1]
        n_id = b["n_out"][-1][:int(counts[hops])]
        if b.get("rows_live"):     # the rows the builder gathered behind this block ride on its n_id (block_rows)
            n_all = int(counts[hops])
            n_id._dgnn_rows = {(src.data_ptr(), col0, cols, which): out[:(n_all if which == "all" else b["nb"])]
                               for (src, col0, cols, which), out in zip(b["rows_spec"], b["rows_out"])}
            if self._escaped is not None:      # fresh buffers (no reuse ring): the consumer's stream must be recorded on them like on the block's arrays
                self._escaped += list(b["rows_out"])
        return b["nb"], n_id, adjs

    def _sample_regular(self, n_id: torch.Tensor):
        return self._finish_regular(self._start_regular(n_id, background=False))

    def _sample(self, batch: torch.Tensor):
        L, st, p = lib(), stream_ptr(), self.plan
        n_id = batch.to(self.device, torch.int64).contiguous()
        batch_size = n_id.numel()
        if self._regular and batch_size > 0 and ONE_CALL:
            return self._sample_regular(n_id)
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
            plan = GraphPlan(ei, n_all, n_t, parts=(off, src32, self._arange(n_e)))
            plan.edge_rows = e_id.to(torch.int32)   # rows of the scene's edge_attr behind the block's edges (see _finish_regular)
            register_plan(ei, plan)
            if self._escaped is not None:
                self._escaped += [ei, e_id, off, src32, self._iota, n_id_out, plan.edge_rows]
                if self.transposed_plans:
                    self._escaped += list(plan.transposed) + [plan.transposed_edge_rows]
            adjs.append(EdgeIndex(ei, e_id if self.return_e_id else None, (n_all, n_t)))
        check(L.dgnn_khop_reset(ptr(n_id), n_id.numel(), ptr(self._pos), st), "dgnn_khop_reset")
        adjs = adjs[0] if len(adjs) == 1 else adjs[::-1]
        return batch_size, n_id, adjs



def block_rows(n_id, src, col0, cols, which="all"):
    """The rows src[ids, col0:col0 + cols] of a block (ids = n_id, or its targets for which = "batch") when the block's builder gathered them
    (NeighborSampler.attach_rows), else None -- the caller then indexes `src` itself."""
    d = getattr(n_id, "_dgnn_rows", None)
    if not d or not isinstance(src, torch.Tensor):
        return None
    return d.get((src.data_ptr(), int(col0), int(cols), which))
