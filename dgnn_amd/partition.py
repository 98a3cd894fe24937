"""Multi-GPU inference on one large scene: spatial partition + per-round halo exchange (RCCL over xGMI).

The reference is single-process / single-GPU; large scenes are handled there by k-hop neighbour
sampling with heavy recomputation (inference_batch_layer, surfaceNetStaticEdgeFilters.py:232-275).
On an 8 x MI355X node the scene is instead cut into one part per rank:

* every rank OWNS a set of tets (recursive coordinate bisection of tet centroids), all 4 in-edges of
  each owned tet and those edges' feature rows -- edge features are static inputs, so no edge data
  ever moves;
* per message-passing round a rank needs the layer input of the remote sources of its in-edges: its
  HALO.  Before conv layer l >= 1 every rank sends the rows other ranks need (packed by one gather
  kernel) and receives its halo rows straight into the tail of its activation buffer
  [n_own + n_halo, C] with one grouped send/recv (torch.distributed.batch_isend_irecv -> grouped
  ncclSend/ncclRecv on RCCL; every pair of GPUs has a direct xGMI link, so this is a single-hop
  neighbour exchange, not a ring).  Layer 0 needs no exchange: halo input features are part of the
  rank's input.  BatchNorm(eval) is per-channel affine -> no communication.
* local ids: owned tets first -- INTERIOR tets (no remote source, not needed by any peer) then BOUNDARY
  tets, ascending global id inside each group -- then halo rows grouped by owner rank (ascending global
  id inside a group) so each peer's rows land contiguously.  The local edge list is grouped by
  destination and keeps the global edge order inside a destination, therefore every destination sums
  its 4 messages in the same order as the single-GPU run and the result is bit-identical to it (and the
  plan builder's "already grouped" fast path applies).
* overlap: the halo exchange of layer l's output runs on its own HIP stream while the interior tets of
  layer l+1 are computed (they read owned rows only); the boundary tets follow once the halo has landed,
  and their outputs are exactly what the next exchange sends.

`layer_fn(i, h, plan_like) -> [n_own, C_out]` and `decoder_fn(h)` are injected, so the same
orchestration runs with the HIP layers on GPUs (product) and with the CPU oracle under gloo (tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np
import torch


def rcb_partition(centroids: np.ndarray, parts: int) -> np.ndarray:
    """Recursive coordinate bisection: splits along the longest axis into (almost) equal counts.
    Returns part id per tet, int32 [N].  `parts` may be any positive integer."""
    n = centroids.shape[0]
    out = np.zeros(n, dtype=np.int32)

    def rec(idx, lo, k):
        if k == 1:
            out[idx] = lo
            return
        c = centroids[idx]
        axis = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
        k_left = k // 2
        n_left = (len(idx) * k_left) // k
        order = np.argsort(c[:, axis], kind="stable")
        rec(idx[order[:n_left]], lo, k_left)
        rec(idx[order[n_left:]], lo + k_left, k - k_left)

    rec(np.arange(n), 0, parts)
    return out


@dataclass
class LocalPart:
    """Index structures of one rank (host arrays)."""
    rank: int
    world: int
    n_total: int
    own_gid: np.ndarray          # int64 [n_own] global ids of owned tets: interior (ascending) then boundary (ascending)
    n_interior: int              # owned tets that neither read a halo row nor are sent to a peer
    halo_gid: np.ndarray         # int64 [n_halo] global ids of halo rows, grouped by owner
    edge_index: np.ndarray       # int64 [2, E_loc] local ids, grouped by dst (ascending); dst < n_own, src < n_own + n_halo
    edge_gid: np.ndarray         # int64 [E_loc] global edge positions (rows of the global edge_attr), ascending per dst
    send_idx: np.ndarray         # int64 [n_send] local ids of owned rows to send, grouped by destination rank
    send_counts: List[int]       # rows sent to each rank
    recv_counts: List[int]       # rows received from each rank (== sizes of the halo groups)
    # ring form (build_ring_part): halo_gid = the cells 1, 2, ... hops from the owned set, ring after ring; nothing is sent or received
    ring_counts: Optional[List[int]] = None

    @property
    def n_own(self):
        return int(self.own_gid.shape[0])

    @property
    def n_halo(self):
        return int(self.halo_gid.shape[0])


def build_local_part(edge_index: np.ndarray, part: np.ndarray, rank: int, world: int) -> LocalPart:
    """edge_index: int64 [2,E] global (row0 = src, row1 = dst); part: owner rank per tet."""
    src, dst = np.asarray(edge_index[0], np.int64), np.asarray(edge_index[1], np.int64)
    n = part.shape[0]
    p_src, p_dst = part[src], part[dst]
    # cut edges s -> d (owner(s) != owner(d)): s is needed by owner(d).  unique (needer, owner, s) triples
    cut = p_src != p_dst
    key = np.unique((p_dst[cut].astype(np.int64) * world + p_src[cut]) * n + src[cut])
    needer, owner, gid = key // (n * world), (key // n) % world, key % n
    mine = needer == rank                       # rows I receive, sorted by (owner, gid)
    halo_gid = gid[mine]
    recv_counts = np.bincount(owner[mine], minlength=world).astype(int).tolist()
    theirs = owner == rank                      # rows I send, sorted by (needer, gid)
    send_gid = gid[theirs]
    send_counts = np.bincount(needer[theirs], minlength=world).astype(int).tolist()
    # owned tets: interior first, then boundary = reads a halo row (cut in-edge) or is sent to a peer (cut out-edge)
    boundary = np.zeros(n, dtype=bool)
    boundary[dst[cut & (p_dst == rank)]] = True
    boundary[send_gid] = True
    owned = part == rank
    own_gid = np.concatenate([np.nonzero(owned & ~boundary)[0], np.nonzero(owned & boundary)[0]]).astype(np.int64)
    n_interior = int(np.count_nonzero(owned & ~boundary))
    # global -> local id map (only for ids this rank touches)
    loc = np.full(n, -1, dtype=np.int64)
    loc[own_gid] = np.arange(own_gid.shape[0])
    send_idx = loc[send_gid]
    loc[halo_gid] = own_gid.shape[0] + np.arange(halo_gid.shape[0])
    e_sel = np.nonzero(p_dst == rank)[0]        # in-edges of owned tets, ascending global position
    e_sel = e_sel[np.argsort(loc[dst[e_sel]], kind="stable")]   # grouped by local destination, global order inside
    e_loc = np.stack([loc[src[e_sel]], loc[dst[e_sel]]])
    assert (e_loc >= 0).all()
    return LocalPart(rank, world, n, own_gid, n_interior, halo_gid, e_loc, e_sel.astype(np.int64), send_idx, send_counts,
                     recv_counts)


def build_ring_part(edge_index: np.ndarray, part: np.ndarray, rank: int, world: int, hops: int) -> LocalPart:
    """The part of `rank` with `hops` rings of halo cells and NO exchange: ring k = the cells k hops (along in-edges) from the owned set.  A stack of
    `hops` conv layers computes layer l for the owned cells and rings 1 .. hops-1-l (LocalPart.ring_dst) -- the reference's k-hop recomputation
    (learning/surfaceNetStaticEdgeFilters.py:232-275) applied to a whole part: every rank is independent of the others during a forward.
    Local ids: owned cells (ascending global id: the scene's locality order carries over), ring 1, ring 2, ... (each ascending); `edge_index` holds
    the in-edges of the owned cells and of rings 1 .. hops-1 (the cells some layer computes), grouped by local destination."""
    src, dst = np.asarray(edge_index[0], np.int64), np.asarray(edge_index[1], np.int64)
    n = part.shape[0]
    have = part == rank
    own_gid = np.nonzero(have)[0].astype(np.int64)
    have = have.copy()
    rings = []
    for _ in range(hops):
        new = np.unique(src[have[dst] & ~have[src]])
        rings.append(new.astype(np.int64))
        have[new] = True
    halo_gid = np.concatenate(rings) if rings else np.zeros(0, np.int64)
    loc = np.full(n, -1, dtype=np.int64)
    loc[own_gid] = np.arange(own_gid.shape[0])
    loc[halo_gid] = own_gid.shape[0] + np.arange(halo_gid.shape[0])
    n_computed = own_gid.shape[0] + sum(r.shape[0] for r in rings[:-1])       # the outermost ring is only read
    ld = loc[dst]
    e_sel = np.nonzero((ld >= 0) & (ld < n_computed))[0]
    e_sel = e_sel[np.argsort(ld[e_sel], kind="stable")]
    e_loc = np.stack([loc[src[e_sel]], ld[e_sel]])
    assert (e_loc >= 0).all()
    zeros = [0] * world
    lp = LocalPart(rank, world, n, own_gid, int(own_gid.shape[0]), halo_gid, e_loc, e_sel.astype(np.int64), np.zeros(0, np.int64), zeros, list(zeros),
                   ring_counts=[int(r.shape[0]) for r in rings])
    lp.trusted_grouped = bool(e_loc.shape[1] == 0 or (np.all(np.diff(e_loc[1]) >= 0) and e_loc[1, -1] < n_computed and e_loc.min() >= 0))   # by construction; checked once, here
    return lp


def ring_dst(lp: LocalPart, num_layers: int) -> List[int]:
    """destinations of layer l of a `num_layers`-deep stack on a ring part: owned cells + rings 1 .. num_layers-1-l"""
    assert lp.ring_counts is not None and len(lp.ring_counts) >= num_layers, "the part was built with fewer rings than the model has conv layers"
    return [lp.n_own + sum(lp.ring_counts[:num_layers - 1 - l]) for l in range(num_layers)]


_FAULT_FIRED = [False]


def _fault_first_exchange():
    """Fault injection for tests (tests/test_gpu_multi.py): DGNN_FAULT_FIRST_EXCHANGE = "all" or a rank as in $RANK makes the FIRST exchange this process
    starts raise before anything is posted -- what bench.py's agreed fall-back to host staging answers.  Use "all": a rank that alone stays out of an
    exchange leaves its peers waiting for its rows (the fall-back agrees AFTER the failed step; it cannot recall posted receives)."""
    import os
    v = os.environ.get("DGNN_FAULT_FIRST_EXCHANGE")
    if not v or _FAULT_FIRED[0]:
        return
    if v == "all" or v == os.environ.get("RANK"):
        _FAULT_FIRED[0] = True
        raise RuntimeError("injected fault (DGNN_FAULT_FIRST_EXCHANGE): the first halo exchange of this process")


class HaloExchange:
    """Per-round exchange of boundary rows.  `pack(h, idx)` gathers rows (HIP gather kernel on GPU).

    start(h) posts the exchange, wait() makes the caller's stream see the received rows; on a GPU the sends/receives are
    issued on a side stream so kernels queued between start() and wait() overlap with the transfer.  __call__ = both."""

    def __init__(self, lp: LocalPart, device, pack: Optional[Callable] = None, group=None, via_host: Optional[bool] = None):
        self.lp = lp
        self.device = torch.device(device)
        self.group = group
        if via_host is None:
            # RCCL moves device buffers directly; any other backend (gloo: bring-up / single-GPU validation runs with
            # several ranks on one device) gets the rows staged through host memory.  Transport only -- never compute.
            import torch.distributed as dist
            via_host = (self.device.type == "cuda" and dist.is_available() and dist.is_initialized()
                        and dist.get_backend(group) != "nccl")
        self.via_host = bool(via_host)
        self.send_idx = torch.from_numpy(lp.send_idx).to(device)
        self.send_idx32 = self.send_idx.to(torch.int32)
        self.pack = pack
        self.n_own = lp.n_own
        # a rank with nothing to send or receive skips the collective (a single-rank part has no halo; a part may also list
        # itself as its own peer -- RCCL allows self send/recv inside a group -- which the RCCL smoke test uses)
        self.active = (sum(lp.send_counts) + sum(lp.recv_counts)) > 0
        self.stream = torch.cuda.Stream(device=self.device) if (self.device.type == "cuda" and self.active) else None
        self._pending = None
        # The exchange as LIBRARY calls (csrc/halo.hip: pack kernel + one RCCL send / recv group on the library's side stream, an event handed
        # back) instead of torch.distributed ops issued one by one from Python -- under the RCCL backend, unless DGNN_NATIVE_HALO=0.  The
        # communicator is the library's own: rank 0's unique id travels over the process group, every rank of the group joins (also ranks without
        # a halo: communicator creation is collective).  Anything that goes wrong here leaves the torch.distributed transport in place.
        self._native = None
        self._native_bufs = {}
        import os
        # DGNN_NATIVE_HALO=force (tests): the library's communicator is ATTEMPTED under any backend, also when this object was built for host staging -- the
        # agreement protocol of _init_native then runs over that backend (gloo on a one-GPU box) and its failure branches can be exercised there.
        force = os.environ.get("DGNN_NATIVE_HALO", "1") == "force"
        self.native_attempt = None          # None: not attempted; else "ok" or the reason the group agreed to stay on torch.distributed
        if self.device.type == "cuda" and (not self.via_host or force) and os.environ.get("DGNN_NATIVE_HALO", "1") != "0":
            import torch.distributed as dist
            # (ring parts -- build_ring_part -- never exchange: no communicator is made for them, on any rank)
            if dist.is_available() and dist.is_initialized() and (dist.get_backend(group) == "nccl" or force) and lp.ring_counts is None \
                    and (self.active or dist.get_world_size(group) > 1):
                try:
                    self._init_native(group)
                    self.native_attempt = "ok"
                except Exception as e:  # noqa: BLE001
                    import sys
                    sys.stderr.write("dgnn_amd: library halo exchange unavailable (%s); torch.distributed transport is used\n" % e)
                    self._native = None
                    self.native_attempt = str(e)

    def _init_native(self, group):
        """Creates the library's communicator + halo plan.  COLLECTIVE over `group`, and every decision on the way is AGREED ON by all ranks
        (a rank that failed locally -- no RCCL to open, no unique id on rank 0 -- used to fall back alone while its peers blocked in the
        broadcast / ncclCommInitRank, or later mixed the two transports): (1) all-reduce(MIN) of "RCCL opens here" and, from rank 0, "the unique id
        exists"; only if every rank says yes does anyone enter the broadcast and `dgnn_comm_create`; (2) all-reduce(MIN) of the local outcome of
        communicator + plan creation; on a zero every rank destroys what it made and raises, so the whole group stays on torch.distributed."""
        import ctypes as C
        import torch.distributed as dist
        from ._lib import check, lib
        L = lib()
        lp = self.lp
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def agree(flag: bool) -> bool:
            if world == 1:
                return bool(flag)
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(t.item())

        with torch.cuda.device(self.device):
            ok = bool(L.dgnn_rccl_available())
            uid = torch.zeros(128, dtype=torch.uint8)
            if ok and rank == 0:
                buf = (C.c_ubyte * 128)()
                ok = L.dgnn_comm_unique_id(buf) == 0
                if ok:
                    uid = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
            if not agree(ok):
                raise RuntimeError("RCCL (or rank 0's unique id) is unavailable on at least one rank of the group")
            if world > 1:
                t = uid.to(self.device)
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                uid = t.cpu()
            idb = (C.c_ubyte * 128)(*uid.tolist())
            comm, plan = C.c_void_p(), C.c_void_p()
            err = None
            try:
                check(L.dgnn_comm_create(idb, rank, world, C.byref(comm)), "dgnn_comm_create")
                sc = (C.c_int64 * world)(*[int(v) for v in lp.send_counts])
                rc = (C.c_int64 * world)(*[int(v) for v in lp.recv_counts])
                check(L.dgnn_halo_plan_create(rank, world, lp.n_own, C.c_void_p(self.send_idx32.data_ptr()) if self.send_idx32.numel() else None, sc, rc,
                                              C.byref(plan)), "dgnn_halo_plan_create")
            except Exception as e:  # noqa: BLE001 -- reported after the agreement below
                err = e
            if not agree(err is None):
                if plan:
                    L.dgnn_halo_plan_destroy(plan)
                if comm:
                    L.dgnn_comm_destroy(comm)
                raise RuntimeError("communicator / halo plan creation failed on at least one rank%s" % ("" if err is None else " (here: %s)" % err))
        self._native = (L, comm, plan)

    def __del__(self):
        nat = getattr(self, "_native", None)
        if nat is not None:
            try:
                L, comm, plan = nat
                L.dgnn_halo_plan_destroy(plan)
                L.dgnn_comm_destroy(comm)
            except Exception:  # noqa: BLE001 -- interpreter teardown
                pass
            self._native = None

    def _post(self, h_full, send):
        import torch.distributed as dist
        lp = self.lp
        if self.via_host:
            recv_dev, send_dev = h_full[self.n_own:], send
            send = send_dev.cpu()                                   # synchronises the host with the current (side) stream
            h_full = torch.empty((h_full.size(0), h_full.size(1)), dtype=h_full.dtype)   # only the halo tail is used
        ops, so, ro = [], 0, self.n_own
        for peer in range(lp.world):
            ns, nr = lp.send_counts[peer], lp.recv_counts[peer]
            if nr:
                ops.append(dist.P2POp(dist.irecv, h_full[ro:ro + nr], peer, group=self.group))
            if ns:
                ops.append(dist.P2POp(dist.isend, send[so:so + ns], peer, group=self.group))
            so += ns
            ro += nr
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if self.via_host:
            recv_dev.copy_(h_full[self.n_own:].to(recv_dev.device, non_blocking=False))

    def start(self, h_full: torch.Tensor) -> None:
        """h_full [n_own + n_halo, C] with the rows to send valid: begins filling the halo rows in place."""
        if not self.active:
            return
        _fault_first_exchange()
        if self._native is not None:
            from ._lib import check, ptr, stream_ptr
            L, comm, plan = self._native
            esz, c = h_full.element_size(), h_full.size(1)
            key = (c, esz)
            buf = self._native_bufs.get(key)
            if buf is None:      # one packed-rows buffer per row shape, reused by every exchange (the next pack is ordered behind the previous wait)
                buf = self._native_bufs[key] = torch.empty(max(1, int(sum(self.lp.send_counts)) * c * esz), dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                check(L.dgnn_halo_exchange_start(plan, comm, ptr(h_full), h_full.stride(0), c, esz, ptr(buf), stream_ptr()), "dgnn_halo_exchange_start")
            self._pending = "native"
            return
        if h_full.dtype == torch.int16:      # unsigned 16-bit rows (dgnn_amd.ops.UROWS): the collective moves them as bf16 bit patterns
            h_full = h_full.view(torch.bfloat16)
        send = self.pack(h_full, self.send_idx32) if self.pack is not None else h_full.index_select(0, self.send_idx)
        if self.stream is None:
            self._post(h_full, send)
            return
        main = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(main)           # packed rows (and the buffer itself) are ready
        with torch.cuda.stream(self.stream):
            self._post(h_full, send)
        send.record_stream(self.stream)
        h_full.record_stream(self.stream)
        self._pending = True

    def wait(self) -> None:
        if self._pending == "native":
            from ._lib import check, stream_ptr
            L, _, plan = self._native
            with torch.cuda.device(self.device):
                check(L.dgnn_halo_exchange_wait(plan, stream_ptr()), "dgnn_halo_exchange_wait")
            self._pending = None
        elif self._pending:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self._pending = None

    def __call__(self, h_full: torch.Tensor) -> torch.Tensor:
        self.start(h_full)
        self.wait()
        return h_full


RING_TRUSTED_PLAN = __import__("os").environ.get("DGNN_RING_TRUSTED_PLAN", "1") != "0"     # 0: a ring part's plan goes through the verified GROUPED builder


class LoopbackExchange(HaloExchange):
    """Timing stand-in for a single-process measurement of one rank's launch chain (tools/bench_partition_rank.py): the packed rows are copied
    into the head of the halo tail on the side stream instead of travelling to a peer.  Not a transport: the halo rows are not the peers' rows."""

    def __init__(self, lp: LocalPart, device, pack: Optional[Callable] = None):
        super().__init__(lp, device, pack=pack, group=None, via_host=False)
        self._native = None

    def _post(self, h_full, send):
        tail = h_full[self.n_own:]
        k = min(send.size(0), tail.size(0))
        if k:
            tail[:k].copy_(send[:k])


def run_partitioned_layers(lp: LocalPart, x_local: torch.Tensor, num_layers: int, layer_fn: Callable, decoder_fn: Callable,
                           exchange: HaloExchange, alloc: Callable, widths) -> torch.Tensor:
    """x_local [n_own + n_halo, F] (halo input rows included).  `layer_fn(i, h, out, b, e)` computes layer i for the owned
    tets [b, e) into out[b:e]; `widths[i]` = output width of layer i.  Returns logits of the owned tets.

    Layer 0 needs no exchange.  For l >= 1 the halo of the layer's input is in flight when the layer starts: interior
    tets first (owned rows only), then wait, then boundary tets -- whose outputs are what the next exchange sends."""
    h = x_local
    n_own, n_int = lp.n_own, lp.n_interior
    for i in range(num_layers):
        last = i == num_layers - 1
        buf = alloc(n_own if last else n_own + lp.n_halo, widths[i])
        if i == 0:
            layer_fn(i, h, buf, 0, n_own)
        else:
            if n_int:
                layer_fn(i, h, buf, 0, n_int)
            exchange.wait()
            if n_own > n_int:
                layer_fn(i, h, buf, n_int, n_own)
        if not last:
            exchange.start(buf)
        h = buf
    return decoder_fn(h[:n_own])


class PartitionedScene:
    """Rank-local scene resident on one GPU + the HIP-side orchestration of a partitioned forward."""

    def __init__(self, lp: LocalPart, x_local: torch.Tensor, edge_attr_local: torch.Tensor, device):
        from . import ops
        from .graph import GraphPlan
        self.lp = lp
        self.device = device
        self.n_total, self.n_own, self.n_halo = lp.n_total, lp.n_own, lp.n_halo
        self.x_local = x_local.to(device)
        self.edge_attr = edge_attr_local.to(device)
        self.edge_index = torch.from_numpy(lp.edge_index).to(device)
        self.exchange = HaloExchange(lp, device, pack=ops.gather_rows)
        self._GraphPlan = GraphPlan
        self.plan = None
        self._xe_stripped = None

    @staticmethod
    def build_synthetic(points: int, seed: int, rank: int, world: int, device, keep_global: bool = False, loader_order: bool = True,
                        halo: str = "exchange", hops: int = 4) -> "PartitionedScene":
        """The seeded Delaunay scene of bench.py cut into `world` parts.  Rank 0 runs the tetrahedralisation and the
        coordinate bisection ONCE and broadcasts the adjacency column and the owner map (two int32 arrays) when a process
        group exists -- every rank repeating scipy.spatial.Delaunay on the whole scene costs minutes at 10M tets; without
        a process group (tests, world == 1) the rank builds them itself.  Each rank then derives its own index structures
        and materialises only the feature rows it needs (hashed N(0,1) values, identical across ranks for shared rows).
        `loader_order`: the generated scene is first relabelled the way the package's loader relabels every scene it reads (Morton order).
        `keep_global`: the whole scene's adjacency column (int32 [4N]) stays attached as `scene.global_dst` (bench.py's check at N > 1 runs
        the same scene as one graph on rank 0)."""
        import torch.distributed as dist
        from .synthetic import delaunay_tet_graph, hashed_normal
        shared = world > 1 and dist.is_available() and dist.is_initialized()
        if not shared or rank == 0:
            adj, cent, _ = delaunay_tet_graph(points, seed)
            if loader_order:
                # cells numbered as the package's loader leaves a scene (ingest-time Morton order, processing/reorder.py): owned cells keep ascending
                # global ids inside a part, so every part inherits the locality
                from .synthetic import loader_cell_order
                adj, cent, _ = loader_cell_order(adj, cent)
            part = rcb_partition(cent, world)
            dst = np.ascontiguousarray(adj[:, 1], dtype=np.int32)
            del adj, cent
        if shared:
            on_gpu = dist.get_backend() == "nccl"
            tdev = torch.device(device) if on_gpu else torch.device("cpu")
            meta = [int(part.shape[0])] if rank == 0 else [None]
            dist.broadcast_object_list(meta, src=0)
            n = meta[0]
            t_dst = torch.from_numpy(dst).to(tdev) if rank == 0 else torch.empty(4 * n, dtype=torch.int32, device=tdev)
            t_part = torch.from_numpy(part.astype(np.int32)).to(tdev) if rank == 0 else torch.empty(n, dtype=torch.int32, device=tdev)
            dist.broadcast(t_dst, src=0)
            dist.broadcast(t_part, src=0)
            dst, part = t_dst.cpu().numpy(), t_part.cpu().numpy()
            del t_dst, t_part
        n = part.shape[0]
        ei = np.empty((2, 4 * n), dtype=np.int64)
        ei[0] = np.repeat(np.arange(n, dtype=np.int64), 4)   # the reference layout: row 4t+r = (t, r-th neighbour)
        ei[1] = dst
        # halo "recompute": `hops` rings of halo cells resident, every layer recomputed on the rings later layers read, no exchange (build_ring_part);
        # "exchange": one ring, its rows exchanged between the layers (RCCL) -- the form the partitioned BACKWARD uses
        lp = build_ring_part(ei, part, rank, world, hops) if halo == "recompute" else build_local_part(ei, part, rank, world)
        del ei
        rows = np.concatenate([lp.own_gid, lp.halo_gid])
        x_local = hashed_normal(rows, 29, seed=1, device=device)
        ea_local = hashed_normal(lp.edge_gid, 20, seed=2, device=device)
        scene = PartitionedScene(lp, x_local, ea_local, device)
        scene.global_dst = np.ascontiguousarray(dst, dtype=np.int32) if keep_global else None
        return scene

    @torch.no_grad()
    def inference_layer(self, net, rebuild_plan: bool = True) -> torch.Tensor:
        """Partitioned equivalent of SurfaceNet.inference_layer: logits [n_own, 2] of the owned tets."""
        n_src = self.n_own + self.n_halo
        x = self.x_local[:, 1:] if net.clf.regularization.cell_type else self.x_local
        x = net._storage_input(x) if hasattr(net, "_storage_input") else x
        if net.clf.regularization.edge_type:
            # column 0 is the regularisation column (reference :334-337): strip it once, the kernels want packed rows
            if self._xe_stripped is None:
                self._xe_stripped = self.edge_attr[:, 1:].contiguous()
            xe = self._xe_stripped
        else:
            xe = self.edge_attr

        if self.lp.ring_counts is not None:
            return self._infer_rings(net, x, xe, n_src, rebuild_plan)
        one = self._infer_one_call(net, x, xe, n_src, rebuild_plan)
        if one is not None:
            return one
        if self.plan is None or rebuild_plan:
            self.plan = self._GraphPlan(self.edge_index, n_src, self.n_own, hint=1)  # local list is grouped by destination
        plan = self.plan
        last = net.num_layers - 1
        # the last layer's launches carry the decoder when the model can do that (as the whole-graph inference_layer does: same kernel, so a
        # partitioned scene stays bit-identical to the whole graph); they then write logits [n_own, 2] instead of rows
        fuse = bool(getattr(net, "fuses_decoder", lambda i: False)(last)) and net._fusable_rows(x, last)

        def layer_fn(i, h, out, b, e):
            net._eval_layers(h, self.n_own, xe, [plan] * net.num_layers, False, only=i, out=out, rows=(b, e), decode=fuse and i == last)

        widths = list(net.clf.model.convs)
        if fuse:
            widths[last] = 2

        act = getattr(net, "activation_dtype", None)
        n_alloc = [0]

        def alloc(r, c):      # layer i's rows in the format the model stores them in; the decoder-carrying last launch writes fp32 logits
            i = n_alloc[0]
            n_alloc[0] += 1
            logits = fuse and i == last
            dt = torch.float32 if logits else (act(i, x.dtype) if act is not None else getattr(net, "storage_dtype", torch.float32))
            return torch.empty((r, c), dtype=dt, device=self.device)
        return run_partitioned_layers(self.lp, x, net.num_layers, layer_fn, (lambda lg: lg) if fuse else net._eval_decoder, self.exchange, alloc,
                                      widths=widths)


def _partitioned_scene_infer_one_call(self, net, x, xe, n_src, rebuild_plan):
    """The step as ONE library call (dgnn_static_infer_partitioned_fwd: plan, layer 0, [interior | wait | boundary] + exchange start per later layer,
    decoder-carrying last launches) when the model's configuration has that form and the exchange is the library's own (RCCL; or there is none to
    do: a single-rank part).  A strong-scaling shard is otherwise host-bound: 0.58 ms of interpreter time per step in front of 0.2 ms of GPU work
    at 1/8 of the 1M-tet scene (tools/bench_partition_rank.py).  None: the per-layer chain below runs (any other transport, bf16 storage, hooks)."""
    from . import ops
    self.used_one_call = False
    if not getattr(self, "one_call", True) or getattr(net, "_one_call_tables", None) is None:
        return None
    ex = self.exchange
    if type(ex) is not HaloExchange or (ex.active and ex._native is None) or ex._pending:
        return None
    tabs = net._one_call_tables(x, xe) if net.storage_dtype == torch.float32 else None
    if tabs is None:
        return None
    layers, decoder, prepared, with_dec, cache = tabs
    if with_dec and not net._fusable_rows(x, net.num_layers - 1):      # (the per-layer chain runs layer and decoder apart there: stay identical to it)
        return None
    halo = comm = sbuf = None
    if ex.active:
        _, comm, halo = ex._native
        maxw = max(int(l[2].size(0)) for l in layers)
        nbytes = max(1, int(sum(self.lp.send_counts)) * maxw * 4)
        sbuf = getattr(self, "_one_call_sbuf", None)
        if sbuf is None or sbuf.numel() < nbytes:
            sbuf = self._one_call_sbuf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
    build = self.plan is None or rebuild_plan
    parts = None if build else (self.plan.rowptr, self.plan.src, self.plan.eid)
    out = ops.static_infer_partitioned_fwd(x, xe, self.edge_index, parts, self.n_own, self.lp.n_interior, layers, decoder, prepared, halo=halo, comm=comm,
                                           send_buf=sbuf, fuse_decoder=with_dec, cache=cache)
    if out is None:
        return None
    if build:
        self.plan = self._GraphPlan(self.edge_index, n_src, self.n_own, hint=1, parts=out[1])
    self.used_one_call = True
    return out[0]


PartitionedScene._infer_one_call = _partitioned_scene_infer_one_call


def _partitioned_scene_infer_rings(self, net, x, xe, n_src, rebuild_plan):
    """Forward of a ring part (build_ring_part): layer l over the owned cells and the rings that later layers still read; no exchange.  One library call
    (dgnn_static_infer_rings_fwd) where the model's configuration has that form, else the per-layer entry points over the same destination prefixes."""
    from . import ops
    lp, L = self.lp, net.num_layers
    nd = ring_dst(lp, L)
    self.used_one_call = False
    # A part built with MORE rings than the model has conv layers (build_synthetic's default hops=4 under a 3-layer net) also lists the in-edges of
    # rings this stack never computes.  The local list is grouped by ascending destination, so the edges of the cells layer 0 computes are a PREFIX
    # of it: the plan (n_key = nd[0]) is built over that prefix only -- surplus edges would otherwise be clamped into the last row by the trusted
    # builder.
    ei = self.edge_index
    if len(lp.ring_counts) > L:
        n_e = int(np.searchsorted(lp.edge_index[1], nd[0], side="left"))
        ei, xe = ei[:, :n_e], xe[:n_e]
        self._ring_ei = ei         # (a plan holds its edge list weakly)
    if self.plan is not None and (self.plan.n_dst != nd[0] or self.plan.E != ei.size(1)):
        self.plan = None           # (another model depth was run on this part before)
    build = self.plan is None or rebuild_plan
    last = L - 1
    fuse = bool(getattr(net, "fuses_decoder", lambda i: False)(last)) and net._fusable_rows(x, last)
    tabs = net._one_call_tables(x, xe) if (getattr(self, "one_call", True) and getattr(net, "_one_call_tables", None) is not None) else None
    if tabs is not None and (not tabs[3] or fuse):
        layers, decoder, prepared, with_dec, cache = tabs
        parts = None if build else (self.plan.rowptr, self.plan.src, self.plan.eid)
        # (the local list was laid out grouped by destination by build_ring_part: its plan is ONE launch, nothing to verify -- lp.trusted_grouped)
        hint = ops.PLAN_HINT_GROUPED_TRUSTED if (getattr(lp, "trusted_grouped", False) and RING_TRUSTED_PLAN) else ops.PLAN_HINT_GROUPED
        if net.storage_dtype == torch.bfloat16:
            out = ops.static_infer_rings_fwd_bf16(x, xe, ei, parts, nd, layers, decoder, hint=hint, cache=cache)
        else:
            out = ops.static_infer_rings_fwd(x, xe, ei, parts, nd, layers, decoder, prepared, hint=hint, fuse_decoder=with_dec, cache=cache)
        if out is not None:
            if build:
                self.plan = self._GraphPlan(ei, n_src, nd[0], hint=1, parts=out[1])
            self.used_one_call = True
            return out[0]
    if build:
        self.plan = self._GraphPlan(ei, n_src, nd[0], hint=1)
    plan = self.plan
    act = getattr(net, "activation_dtype", None)
    h = x
    for i in range(L):
        dec = fuse and i == last
        dt = torch.float32 if dec else (act(i, x.dtype) if act is not None else getattr(net, "storage_dtype", torch.float32))
        out = torch.empty((nd[i], 2 if dec else net.clf.model.convs[i]), dtype=dt, device=self.device)
        h = net._eval_layers(h, nd[i], xe, [plan] * L, False, only=i, out=out, rows=(0, nd[i]), decode=dec)
    return h if fuse else net._eval_decoder(h[:self.n_own])


PartitionedScene._infer_rings = _partitioned_scene_infer_rings


def build_self_halo_part(edge_index: np.ndarray, n: int, remote: np.ndarray) -> LocalPart:
    """A single-rank part that is ITS OWN PEER (RCCL allows self send / recv inside a group): the cells `remote` are ALSO held as halo rows -- every
    edge leaving one of them reads the halo copy, which the exchange fills from the owned row.  Its partitioned forward must equal the whole
    scene's bit for bit, so the full RCCL path (plan of the local graph, interior / boundary launches, the library's exchange, decoder) can be
    validated on one GPU (tests/test_gpu_infer.py) and timed (tools/bench_partition_rank.py)."""
    src, dst = np.asarray(edge_index[0], np.int64), np.asarray(edge_index[1], np.int64)
    remote = np.unique(np.asarray(remote, np.int64))
    is_remote = np.zeros(n, dtype=bool)
    is_remote[remote] = True
    cut = is_remote[src]
    boundary = is_remote.copy()
    boundary[dst[cut]] = True
    own_gid = np.concatenate([np.nonzero(~boundary)[0], np.nonzero(boundary)[0]]).astype(np.int64)
    loc = np.empty(n, dtype=np.int64)
    loc[own_gid] = np.arange(n)
    slot = np.full(n, -1, dtype=np.int64)
    slot[remote] = n + np.arange(remote.shape[0])
    e_sel = np.argsort(loc[dst], kind="stable")
    e_src = np.where(cut[e_sel], slot[src[e_sel]], loc[src[e_sel]])
    k = int(remote.shape[0])
    return LocalPart(0, 1, n, own_gid, int(np.count_nonzero(~boundary)), remote.copy(), np.stack([e_src, loc[dst[e_sel]]]), e_sel.astype(np.int64),
                     loc[remote], [k], [k])


def _partitioned_scene_train_forward(self, net, group=None):
    """Train-mode forward of `net` (the Static SurfaceNet) over this rank's part of the scene -> logits [n_own, out] of the owned cells with a
    backward (halo gradients return to their owners, BatchNorm statistics span the scene): see partitioned_train_forward."""
    x = self.x_local[:, 1:] if net.clf.regularization.cell_type else self.x_local
    return partitioned_train_forward(self.lp, x.contiguous() if x.stride(1) != 1 else x, self.edge_attr, self.edge_index, [blk[0] for blk in net.convs],
                                     [getattr(blk[1], "module", None) if len(blk) > 1 else None for blk in net.convs],
                                     net.decoder if net.clf.model.decoder else None, self.exchange, group)


PartitionedScene.train_forward = _partitioned_scene_train_forward


def _needs_host_staging(t: torch.Tensor, group) -> bool:
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _contiguous_strides(shape):
    st, acc = [], 1
    for d in reversed(tuple(shape)):
        st.append(acc)
        acc *= int(d)
    return tuple(reversed(st))


def allreduce_gradients(model: torch.nn.Module, group=None, average: bool = True) -> None:
    """Data-parallel training step helper (BASELINE config 5: one scene shard per GPU, weight replicas):
    ONE collective over the flat fp32 gradient (0.4 MB for the shipped widths, 6.6 MB for [128..1024]) --
    latency-bound, so everything is bucketed into a single ncclAllReduce on RCCL.  BatchNorm statistics stay
    per rank (the reference has a single rank; SyncBN would change its numerics).  Called by Trainer.train between
    loss.backward() and optimizer.step() (learning/runModel.py:279-282).  Parameters without a gradient on this rank (a
    layer the rank's batch never reached) contribute zeros, so every rank reduces the same buffer layout.  Under a
    non-RCCL backend (gloo bring-up runs) GPU gradients are staged through host memory."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in params])
    if _needs_host_staging(flat, group):
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat = host.to(flat.device)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    # the reduced gradients are handed over as views of the flat buffer (one view op per parameter; a copy_ per parameter was 34 launches a step)
    off = 0
    for p in params:
        n = p.numel()
        if p.dtype == torch.float32:
            p.grad = torch.as_strided(flat, tuple(p.shape), tuple(p.stride()) if p.is_contiguous() else _contiguous_strides(p.shape), off)
        else:
            p.grad = flat[off:off + n].view_as(p).to(p.dtype)
        off += n


def broadcast_parameters(model: torch.nn.Module, group=None, src: int = 0) -> None:
    """Replicas start from rank `src`'s weights and buffers (one flat broadcast per dtype class)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    ts = [t for t in list(model.parameters()) + list(model.buffers()) if t.is_floating_point()]
    if not ts:
        return
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in ts])
    if _needs_host_staging(flat, group):
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat = host.to(flat.device)
    else:
        dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
            off += n


# ---- backward of a partitioned single scene (SURVEY 8e "Backward mirrors it"; round 4) ------------------------------------------------------
# Training ONE scene that is cut across ranks (the reference trains on sampled blocks of a resident scene, learning/runModel.py:264-282; a scene that does
# not fit one GPU has no reference counterpart -- this is the partitioned forward's mirror image):
#   * the halo exchange gets a backward: the gradient of the halo rows goes back to their owners and is ADDED to the gradient of the rows they sent;
#   * BatchNorm in training mode takes its batch statistics over the WHOLE scene: one [2 C] all-reduce (sum, sum of squares) forward, one [2 C]
#     all-reduce (sum dy, sum dy * x_hat) backward per layer; the running buffers are updated with the global statistics on every rank;
#   * every rank's parameter gradients are partial sums over its owned cells: ONE flat all-reduce (SUM) at the end of the backward.
# Both pieces are autograd Functions, so the conv layers in between can be the HIP model's (functional.py) or, in the CPU tests, the oracle's modules.


def _rows_reverse_exchange(lp: LocalPart, g_halo: torch.Tensor, group, via_host: bool):
    """sends the gradient rows of the halo (grouped by owner) back to their owners; returns [n_send, C]: the gradient contributions for the rows this
    rank sent, in send_idx order"""
    import torch.distributed as dist
    dev = g_halo.device
    if via_host:
        g_halo = g_halo.cpu()
    back = torch.empty((int(sum(lp.send_counts)), g_halo.size(1)), dtype=g_halo.dtype, device=g_halo.device)
    ops_, so, ro = [], 0, 0
    for peer in range(lp.world):
        ns, nr = lp.send_counts[peer], lp.recv_counts[peer]
        if ns:
            ops_.append(dist.P2POp(dist.irecv, back[so:so + ns], peer, group=group))
        if nr:
            ops_.append(dist.P2POp(dist.isend, g_halo[ro:ro + nr].contiguous(), peer, group=group))
        so += ns
        ro += nr
    if ops_:
        for req in dist.batch_isend_irecv(ops_):
            req.wait()
    return back.to(dev) if via_host else back


class _HaloRows(torch.autograd.Function):
    """h_own [n_own, C] -> [n_own + n_halo, C] with the peers' rows behind the owned ones; backward: halo gradients return to their owners"""

    @staticmethod
    def forward(ctx, h_own, exchange):
        lp = exchange.lp
        full = torch.empty((lp.n_own + lp.n_halo, h_own.size(1)), dtype=h_own.dtype, device=h_own.device)
        full[:lp.n_own] = h_own
        exchange(full)
        ctx.exchange = exchange
        return full

    @staticmethod
    def backward(ctx, g_full):
        ex = ctx.exchange
        lp = ex.lp
        g_own = g_full[:lp.n_own].clone()
        if ex.active:
            back = _rows_reverse_exchange(lp, g_full[lp.n_own:], ex.group, ex.via_host)
            g_own.index_add_(0, ex.send_idx, back)       # a row sent to several peers collects every peer's contribution
        return g_own, None


def halo_rows(h_own: torch.Tensor, exchange: HaloExchange) -> torch.Tensor:
    """differentiable halo exchange (see _HaloRows)"""
    return _HaloRows.apply(h_own, exchange)


class _SceneBatchNorm(torch.autograd.Function):
    """BatchNorm1d in training mode over a scene that is cut across ranks: statistics over ALL ranks' rows (biased variance for the normalisation,
    unbiased for the running buffer, as torch.nn.BatchNorm1d), fp64 sums"""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, group, via_host):
        import torch.distributed as dist
        c = x.size(1)
        xs = x.double()
        st = torch.cat([xs.sum(0), (xs * xs).sum(0), torch.full((1,), float(x.size(0)), dtype=torch.float64, device=x.device)])
        st = _all_reduce_sum(st, group, via_host)
        n = st[2 * c].item()
        mean = st[:c] / n
        var = (st[c:2 * c] / n - mean * mean).clamp_min(0.0)
        invstd = torch.rsqrt(var + bn.eps)
        xhat = (xs - mean) * invstd
        if bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                bn.num_batches_tracked.add_(1)
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - mom).add_((mom * mean).to(bn.running_mean.dtype))
                bn.running_var.mul_(1 - mom).add_((mom * var * (n / max(n - 1.0, 1.0))).to(bn.running_var.dtype))
        ctx.save_for_backward(xhat, gamma, invstd)
        ctx.cfg = (group, via_host, n)
        return (xhat * gamma.double() + beta.double()).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xhat, gamma, invstd = ctx.saved_tensors
        group, via_host, n = ctx.cfg
        c = xhat.size(1)
        dyd = dy.double()
        loc = torch.cat([dyd.sum(0), (dyd * xhat).sum(0)])
        glob = _all_reduce_sum(loc.clone(), group, via_host)
        dx = (gamma.double() * invstd) * (dyd - glob[:c] / n - xhat * (glob[c:] / n))
        # parameter gradients stay LOCAL partial sums: the flat all-reduce at the end of the step adds them up like every other parameter's
        return dx.to(dy.dtype), loc[c:].to(gamma.dtype), loc[:c].to(gamma.dtype), None, None, None


def _all_reduce_sum(t: torch.Tensor, group, via_host: bool) -> torch.Tensor:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    if via_host and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def scene_batch_norm(x: torch.Tensor, bn: torch.nn.BatchNorm1d, group=None, via_host: bool = False) -> torch.Tensor:
    """training-mode BatchNorm1d with scene-wide statistics (see _SceneBatchNorm); eval mode is the module itself (no communication)"""
    if not bn.training:
        return torch.nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    return _SceneBatchNorm.apply(x, bn.weight, bn.bias, bn, group, via_host)


def partitioned_train_forward(lp: LocalPart, x_local: torch.Tensor, edge_attr_local: torch.Tensor, edge_index_local: torch.Tensor, convs, norms,
                              decoder, exchange: HaloExchange, group=None):
    """Train-mode forward of the Static model over a partitioned scene -> logits [n_own, out] of the owned cells, differentiable end to end
    (reference SurfaceNet.forward :196-227 with the whole scene as one block, cut across ranks).
    convs[i]((h_full, h_own), edge_attr, edge_index) is layer i's conv (the HIP model's SAGEConv or the oracle's), norms[i] its BatchNorm1d (or None),
    decoder = the model's decoder Sequential (Linear, norm wrapper | None, ReLU, Linear) or a single Linear.  Layer 0 reads the rank's input rows (its
    halo input features are part of x_local); every later layer exchanges its input's halo first."""
    via_host = bool(exchange.via_host)
    on_gpu = x_local.is_cuda

    def bn_relu(z, bn):
        # GPU: the library's kernels around the collective (functional._SceneBatchNormRelu, BatchNorm + ReLU in one pass each way); CPU (tests): torch ops
        if on_gpu and bn is not None and bn.training and z.dtype == torch.float32:
            from . import functional as Fn
            return Fn.scene_batch_norm_relu(z, bn, lambda t: _all_reduce_sum(t, group, via_host), True)
        return torch.relu(scene_batch_norm(z, bn, group, via_host) if bn is not None else z)

    h_full = x_local
    for i, conv in enumerate(convs):
        if i > 0:
            h_full = halo_rows(h_own, exchange)
        h_own = bn_relu(conv((h_full, h_full[:lp.n_own]), edge_attr_local, edge_index_local), norms[i])
    if decoder is None:
        return h_own
    mods = list(decoder) if isinstance(decoder, torch.nn.Sequential) else [decoder]
    h, k = h_own, 0
    while k < len(mods):
        m = mods[k]
        bn = getattr(m, "module", m)
        if isinstance(bn, torch.nn.BatchNorm1d):
            if k + 1 < len(mods) and isinstance(mods[k + 1], torch.nn.ReLU):      # the decoder's norm -> ReLU pair
                h = bn_relu(h, bn)
                k += 1
            else:
                h = scene_batch_norm(h, bn, group, via_host)
        elif isinstance(m, torch.nn.ReLU):
            h = torch.relu(h)
        elif m is not None:
            if on_gpu:
                from . import functional as Fn
                h = Fn.linear2(h, m.weight, bias=m.bias)
            else:
                h = torch.nn.functional.linear(h, m.weight, m.bias)
        k += 1
    return h


def partitioned_kl_loss(logits: torch.Tensor, gt: torch.Tensor, vol: torch.Tensor, group=None, via_host: bool = False) -> torch.Tensor:
    """the Trainer's volume-weighted KL cell loss (learning/runModel.py:171-209, cell_norm None) of a partitioned scene: sum over ALL ranks' cells of
    w * kl / sum of w; every rank returns the scene's loss, its backward yields this rank's share of the gradient"""
    cell = torch.nn.functional.kl_div(torch.nn.functional.log_softmax(logits, dim=-1), gt[:, :2], reduction="none").sum(1) * vol
    w_all = _all_reduce_sum(vol.double().sum().reshape(1).clone(), group, via_host)[0].to(cell.dtype)
    mine = cell.sum() / w_all
    total = _all_reduce_sum(mine.detach().double().reshape(1).clone(), group, via_host)[0].to(cell.dtype)
    return mine + (total - mine.detach())       # value: the scene's loss; gradient: this rank's term
