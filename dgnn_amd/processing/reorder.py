"""Ingest-time cell locality order (csrc/reorder.hip).

The reference hands the network a scene in the order its CGAL front end wrote the cells (``processing/data.py:434-438`` builds
``edge_index`` straight from ``<scene>_adjacencies.npz``): the 4 neighbours of a cell are tens of thousands of rows apart, so every
neighbour-row gather of the conv layers misses L2.  ``dataLoader.run`` (``processing/data.py`` of this package) therefore relabels
the cells once per scene and remembers the permutation; per-cell results go back to file order where they leave
(``dataLoader.exportScore``, ``generate_mesh.generate``) through ``restore_cell_order``.

``order[i]`` = file-order id of the cell that sits in row i; ``rank`` = its inverse (``rank[order[i]] == i``).  Everything here
is index work on the device; the permutation is deterministic.
"""
from __future__ import annotations

import numpy as np
import torch

from .._lib import check, lib, on_device_of, ptr, stream_ptr

RANK_TAG = "_dgnn_cell_rank"     # attribute the loader puts on its per-cell tensors so that an UNMODIFIED run.py:prepareSample still carries the order


def _i32(n, dev):
    return torch.empty(int(max(n, 1)), dtype=torch.int32, device=dev)


@on_device_of
def centroids_from_3dt(vertices: torch.Tensor, tetrahedra: torch.Tensor, infinite: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """fp32 [n,3] centroids: finite cell i is tetrahedron number (count of finite cells before i) of `_3dt.npz` (generate_mesh.py:78-81),
    an infinite cell sits on its finite neighbour."""
    dev = edge_index.device
    n = infinite.numel()
    v = vertices.to(dev, torch.float32).contiguous()
    t = tetrahedra.to(dev, torch.int32).contiguous()
    inf = infinite.to(dev, torch.int32).contiguous()
    cent = torch.empty((n, 3), dtype=torch.float32, device=dev)
    scratch = _i32(lib().dgnn_cell_centroids_scratch_elems(n), dev)
    check(lib().dgnn_cell_centroids_3dt(ptr(v), v.size(0), ptr(t), t.size(0), ptr(inf), ptr(edge_index), edge_index.stride(0), edge_index.stride(1), n,
                                        ptr(cent), ptr(scratch), stream_ptr()), "dgnn_cell_centroids_3dt", poll=True)
    return cent


@on_device_of
def cell_order_morton(centroids: torch.Tensor):
    """-> (order, rank) int32 [n]: cells sorted (stable) by the 48-bit Morton code of their centroid"""
    c = centroids.to(torch.float32).contiguous()
    n, dev = c.size(0), c.device
    order, rank = _i32(n, dev)[:n], _i32(n, dev)[:n]
    scratch = torch.empty(int(lib().dgnn_cell_order_morton_scratch_elems(n)) // 2 + 1, dtype=torch.int64, device=dev)   # 8-byte aligned
    check(lib().dgnn_cell_order_morton(ptr(c), n, ptr(order), ptr(rank), ptr(scratch), stream_ptr()), "dgnn_cell_order_morton")
    return order, rank


@on_device_of
def cell_order_bfs(edge_index: torch.Tensor, n: int):
    """-> (order, rank) int32 [n] from the adjacency alone (reference layout: row 4t+k of edge_index leaves cell t): breadth-first order"""
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.size(1) != 4 * n:
        raise ValueError("cell_order_bfs needs the reference layout: int64 [2, 4n]")
    dev = edge_index.device
    order, rank = _i32(n, dev)[:n], _i32(n, dev)[:n]
    scratch = _i32(lib().dgnn_cell_order_bfs_scratch_elems(n), dev)
    check(lib().dgnn_cell_order_bfs(ptr(edge_index), edge_index.stride(0), edge_index.stride(1), n, ptr(order), ptr(rank), ptr(scratch), stream_ptr()),
          "dgnn_cell_order_bfs", poll=True)
    return order, rank


@on_device_of
def reorder_edges(edge_index: torch.Tensor, order: torch.Tensor, rank: torch.Tensor):
    """-> (edge_index' int64 [2,4n] = the transposed view of an [4n,2] array, as the reference's loader builds it (data.py:437-438),
    edge_rows int32 [4n] = old row of every new edge row)"""
    n, dev = order.numel(), edge_index.device
    pairs = torch.empty((4 * n, 2), dtype=torch.int64, device=dev)
    rows = _i32(4 * n, dev)[:4 * n]
    check(lib().dgnn_reorder_edges_ref(ptr(edge_index), edge_index.stride(0), edge_index.stride(1), n, ptr(order), ptr(rank), ptr(pairs), ptr(rows),
                                       stream_ptr()), "dgnn_reorder_edges_ref", poll=True)
    return pairs.t(), rows


class CellOrder:
    """A scene's permutation: `order` (row -> file id), `rank` (file id -> row), both int32 on the device, and how it was made."""

    def __init__(self, order, rank, kind):
        self.order, self.rank, self.kind = order, rank, kind

    def to_rows(self, t: torch.Tensor) -> torch.Tensor:
        """per-cell tensor in file order -> row order"""
        return t[self.order.to(t.device).long()]

    def to_file(self, t: torch.Tensor) -> torch.Tensor:
        """per-cell tensor in row order -> file order (where results leave)"""
        return t[self.rank.to(t.device).long()]


def scene_order(edge_index: torch.Tensor, n: int, infinite=None, mfile: str = None, centroids=None, kind: str = "auto"):
    """The order `dataLoader.run` applies: Morton order of the cell centroids when the scene has coordinates (`centroids`, or
    `<scene>_3dt.npz` at `mfile`), else breadth-first order of the adjacency.  kind: auto | morton | bfs.  None when the adjacency is not the reference
    layout (nothing is reordered then)."""
    import os
    if edge_index.dim() != 2 or edge_index.size(1) != 4 * n or n == 0:
        return None
    if kind in ("auto", "morton"):
        if centroids is None and mfile is not None and os.path.isfile(mfile) and infinite is not None:
            m = np.load(mfile)
            if "vertices" in m.files and "tetrahedra" in m.files and len(m["tetrahedra"]) == int((torch.as_tensor(infinite) == 0).sum()):
                centroids = centroids_from_3dt(torch.from_numpy(np.ascontiguousarray(m["vertices"], dtype=np.float32)),
                                               torch.from_numpy(np.ascontiguousarray(m["tetrahedra"]).astype(np.int32)),
                                               torch.as_tensor(infinite), edge_index)
        if centroids is not None:
            return CellOrder(*cell_order_morton(centroids.to(edge_index.device)), "morton")
        if kind == "morton":
            raise ValueError("cell order 'morton' needs coordinates (<scene>_3dt.npz with as many tetrahedra as finite cells)")
    return CellOrder(*cell_order_bfs(edge_index, n), "bfs")


# The permutation of the scenes the loader has relabelled in this process, by the scene's files (`path`, `gtfile`): the per-tensor RANK_TAG does not
# survive `.to()`, `.clone()`, `.cpu()`, `torch.cat`, indexing or pickling, the data object's `path` / `gtfile` strings (run.py:212-213) do.  An entry
# is replaced when the same scene is loaded again and set to None when it is loaded WITHOUT relabelling (so that a stale order is never applied).
# The strings alone do not say that an object's rows ARE in the relabelled order (one built straight from the npz files, a cached dataset, a second
# loader with cell_order off carries the same strings): an entry therefore keeps the relabelled `infinite` flags (1 byte per cell, host memory) and is
# applied only to an object whose own `infinite` equals them -- a positive sign that its rows went through the relabelling loader.  The registry holds
# the _SCENE_ORDERS_MAX most recently loaded scenes.
import collections

_SCENE_ORDERS = collections.OrderedDict()
_SCENE_ORDERS_MAX = 32


def _scene_key(data):
    get = (lambda k: data.get(k)) if isinstance(data, dict) else (lambda k: getattr(data, k, None))
    path, gt = get("path"), get("gtfile")
    if path is None or gt is None:
        return None
    import os
    return os.path.normpath(os.path.join(str(path), str(gt)))


def _flags_u8(v):
    return torch.as_tensor(v).detach().reshape(-1).to("cpu").ne(0).to(torch.uint8)


def register_scene_order(data, co) -> None:
    """called by `dataLoader.run` for every scene it loads (co = None: loaded in file order)"""
    key = _scene_key(data)
    if key is None:
        return
    inf = getattr(data, "infinite", None) if not isinstance(data, dict) else data.get("infinite")
    _SCENE_ORDERS.pop(key, None)
    _SCENE_ORDERS[key] = (co, _flags_u8(inf) if (co is not None and inf is not None) else None)
    while len(_SCENE_ORDERS) > _SCENE_ORDERS_MAX:
        _SCENE_ORDERS.popitem(last=False)


def find_cell_order(data):
    """-> (CellOrder | None, known): the order `data`'s rows are in.  Looked up (1) on the object itself (`cell_order`: the loader, or a dataset
    object that carries the field), (2) on the loader's tagged per-cell tensors (an unmodified run.py:prepareSample hands those on), (3) in the
    registry of scenes this process has loaded, by the object's `path` + `gtfile` -- applied only when the object's `infinite` flags equal the
    relabelled scene's (see _SCENE_ORDERS).  `known` is False when none of the three says anything."""
    said = False
    if (isinstance(data, dict) and "cell_order" in data) or (not isinstance(data, dict) and hasattr(data, "cell_order")):
        co = data["cell_order"] if isinstance(data, dict) else data.cell_order
        if co is not None:
            return co, True
        said = hasattr(data, "run")         # the loader itself: None = its current scene is in file order
        if said:
            return None, True
    for name in ("infinite", "x", "features", "y", "gt"):
        v = getattr(data, name, None) if not isinstance(data, dict) else data.get(name)
        co = getattr(v, RANK_TAG, None) if v is not None else None
        if co is not None:
            return co, True
    key = _scene_key(data)
    if key is not None and key in _SCENE_ORDERS:
        co, flags = _SCENE_ORDERS[key]
        if co is None:
            return None, True
        inf = getattr(data, "infinite", None) if not isinstance(data, dict) else data.get("infinite")
        if inf is not None and flags is not None:
            mine = _flags_u8(inf)
            if mine.numel() == flags.numel() and torch.equal(mine, flags):
                return co, True
        return None, said        # same files, but nothing says these rows were relabelled: left alone
    return None, said


def restore_cell_order(t: torch.Tensor, data) -> torch.Tensor:
    """Per-cell tensor `t` (rows in the order of `data`) back in FILE order; `data` is the loader, a dataset object carrying `cell_order`, one built by
    an unmodified `run.py:prepareSample` (tagged tensors), or any object with the `path` / `gtfile` of a scene this process's loader has loaded
    (the registry: survives `.to()` / `.clone()` / `torch.cat` / pickling of the tensors) -- see `find_cell_order`.  Unchanged when the scene was
    not relabelled.  A permutation whose length is not `t`'s row count raises (a mixed-up scene) instead of scrambling the result."""
    co, _ = find_cell_order(data)
    if co is None:
        return t
    if co.rank.numel() != t.size(0):
        raise RuntimeError("restore_cell_order: the scene's permutation has %d cells, the tensor %d rows" % (co.rank.numel(), t.size(0)))
    return co.to_file(t)
