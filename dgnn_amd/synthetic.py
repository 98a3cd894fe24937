"""Seeded synthetic Delaunay tetrahedron-adjacency graphs in the reference's file layout.

The reference consumes ``<scene>_adjacencies.npz["adjacencies"]``: int32 [E,2], E = 4N, row k =
(src = k//4, dst = neighbour), infinite cells included as ordinary 4-neighbour nodes
(processing/data.py:434-438; layout facts in SURVEY.md section 0).  The external CGAL binary
that writes those files is not in the tree, so benchmarks and tests build the same structure
from ``scipy.spatial.Delaunay``: one node per finite tetrahedron, plus one infinite cell per
convex-hull facet wired to its finite cell and to the 3 infinite cells across the hull edges.
Every node then has out-degree 4 and in-degree 4 and the edge set is symmetric.
"""
from __future__ import annotations

import numpy as np


def delaunay_tet_graph(n_points: int, seed: int = 0):
    """Returns (adjacencies int32 [4N,2], centroids float32 [N,3], n_finite).

    150 000 points / seed 0 gives the BASELINE.md metric graph (N = 1 010 078, E = 4 040 312).
    """
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    pts = rng.random((n_points, 3))
    tri = Delaunay(pts)
    simp = tri.simplices.astype(np.int64)  # [Nf,4]
    nbr = tri.neighbors.astype(np.int64)  # [Nf,4], -1 = hull; nbr[i,k] is opposite vertex k
    nf = simp.shape[0]

    hull_cell, hull_k = np.nonzero(nbr < 0)  # one infinite cell per hull facet
    n_inf = hull_cell.shape[0]
    inf_id = nf + np.arange(n_inf, dtype=np.int64)
    nbr = nbr.copy()
    nbr[hull_cell, hull_k] = inf_id

    # vertices of each hull facet = simplex vertices except the one opposite
    keep = np.ones((n_inf, 4), dtype=bool)
    keep[np.arange(n_inf), hull_k] = False
    fv = simp[hull_cell][keep].reshape(n_inf, 3)
    fv.sort(axis=1)
    # the three hull edges of each facet; each hull edge is shared by exactly two hull facets
    e_a = np.concatenate([fv[:, 0], fv[:, 0], fv[:, 1]])
    e_b = np.concatenate([fv[:, 1], fv[:, 2], fv[:, 2]])
    owner = np.concatenate([np.arange(n_inf)] * 3)
    slot = np.repeat(np.arange(3), n_inf)
    key = e_a * np.int64(n_points) + e_b
    order = np.argsort(key, kind="stable")
    ks = key[order]
    if ks.shape[0] % 2 or not np.array_equal(ks[0::2], ks[1::2]):
        raise RuntimeError("hull is not a closed 2-manifold; cannot wire infinite cells")
    inf_nbr = np.empty((n_inf, 3), dtype=np.int64)
    o0, o1 = order[0::2], order[1::2]
    inf_nbr[owner[o0], slot[o0]] = nf + owner[o1]
    inf_nbr[owner[o1], slot[o1]] = nf + owner[o0]

    n = nf + n_inf
    dst = np.empty((n, 4), dtype=np.int64)
    dst[:nf] = nbr
    dst[nf:, 0] = hull_cell
    dst[nf:, 1:] = inf_nbr
    adj = np.empty((4 * n, 2), dtype=np.int32)
    adj[:, 0] = np.repeat(np.arange(n, dtype=np.int32), 4)
    adj[:, 1] = dst.reshape(-1).astype(np.int32)

    cent = np.empty((n, 3), dtype=np.float32)
    cent[:nf] = pts[simp].mean(axis=1)
    cent[nf:] = pts[fv].mean(axis=1)  # infinite cells sit at their hull facet
    return adj, cent, nf


def check_four_regular(adj: np.ndarray) -> bool:
    """True iff every node has in- and out-degree 4, no self loops, and the edge set is symmetric."""
    n = adj.shape[0] // 4
    src, dst = adj[:, 0].astype(np.int64), adj[:, 1].astype(np.int64)
    if not np.array_equal(src, np.repeat(np.arange(n), 4)):
        return False
    if np.any(src == dst) or np.any(np.bincount(dst, minlength=n) != 4):
        return False
    fwd = np.sort(src * n + dst)
    rev = np.sort(dst * n + src)
    return bool(np.array_equal(fwd, rev))


def loader_cell_order(adj: np.ndarray, cent: np.ndarray):
    """What dgnn_amd.processing.data.dataLoader does to a scene at ingest (processing/reorder.py, Morton order of the cell centroids), restated
    with numpy for scene GENERATORS that run on the host before any GPU work (PartitionedScene.build_synthetic cuts the scene on rank 0's host):
    -> (adjacencies relabelled, reference layout kept, centroids in the new order, order = old id of new cell).  Same keys as
    csrc/reorder.hip (16 bits per axis over the bounding box, stable sort)."""
    n = adj.shape[0] // 4
    c = np.asarray(cent, np.float32)
    lo, hi = c.min(axis=0), c.max(axis=0)
    ext = (hi - lo).astype(np.float32)
    inv = np.where(ext > 0, np.float32(65535.0) / np.where(ext > 0, ext, 1), np.float32(0)).astype(np.float32)
    q = np.clip(((c - lo).astype(np.float32) * inv).astype(np.float32), 0, 65535).astype(np.uint64)
    key = np.zeros(n, np.uint64)
    for a in range(3):
        v = q[:, a]
        v = (v | (v << np.uint64(16))) & np.uint64(0x0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x249249249249)
        key |= v << np.uint64(a)
    order = np.argsort(key, kind="stable")
    rank = np.empty(n, np.int64)
    rank[order] = np.arange(n)
    rows = (order[:, None] * 4 + np.arange(4)[None]).reshape(-1)
    adj2 = np.empty_like(adj)
    adj2[:, 0] = np.repeat(np.arange(n, dtype=adj.dtype), 4)
    adj2[:, 1] = rank[adj[rows, 1].astype(np.int64)].astype(adj.dtype)
    return adj2, np.ascontiguousarray(c[order]), order


def hashed_normal(rows, ncols: int, seed: int = 0, device="cpu"):
    """Deterministic N(0,1) float32 tensor [len(rows), ncols] where entry (i, c) depends only on
    (rows[i], c, seed): splitmix64 hash -> Box-Muller, evaluated with torch integer ops on `device`.
    Lets every rank of a partitioned scene materialise exactly the feature rows it owns (or holds
    as halo) without generating the whole scene, while all ranks agree on shared rows."""
    import torch

    def s64(v):  # python uint64 constant -> wrapped int64
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, k):  # logical shift right on int64
        return (z >> k) & ((1 << (64 - k)) - 1)

    r = torch.as_tensor(np.asarray(rows, dtype=np.int64) if not isinstance(rows, torch.Tensor) else rows, device=device).to(torch.int64)
    z = r.reshape(-1, 1) * ncols + torch.arange(ncols, dtype=torch.int64, device=device).reshape(1, -1)
    z = z + s64(seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019)
    z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    u1 = (lsr(z, 40).to(torch.float32) + 0.5) / float(1 << 24)
    u2 = ((z & 0xFFFFFF).to(torch.float32) + 0.5) / float(1 << 24)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(6.283185307179586 * u2)
