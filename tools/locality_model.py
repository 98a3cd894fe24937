"""Why the wide aggregate reads its rows ~2.3 x: an LRU model of one XCD's L2 over the row requests of k_agg_sr (csrc/wide.hip) on the
metric scene, on the host.  Every cell requests its 4 neighbour rows (the split-row pass never reads the own row); an XCD walks its contiguous eighth of
the cells in index order (the 512 wavefronts of its workgroups sweep it together, 2 048 cells in flight).  The stack distance of every request (distinct
rows touched since the row's last use) gives the miss rate of a fully associative LRU cache of ANY capacity in one pass; a request misses a cache of R
rows iff its stack distance > R.  Cell orders: the loader's Morton order (dgnn_amd/processing/reorder.py), a Hilbert curve of the same keys, the
generator's order, and a breadth-first order.

    python tools/locality_model.py [points]          (CPU only, ~1 minute)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dgnn_amd.synthetic import delaunay_tet_graph, loader_cell_order


def hilbert_keys(cent, bits=16):
    """3-D Hilbert index (Skilling's transpose algorithm) of the quantised centroids"""
    c = np.asarray(cent, np.float64)
    lo, hi = c.min(0), c.max(0)
    X = np.clip(((c - lo) / np.where(hi > lo, hi - lo, 1) * (2 ** bits - 1)), 0, 2 ** bits - 1).astype(np.uint64).T.copy()      # [3, n]
    n = 3
    M = np.uint64(1) << np.uint64(bits - 1)
    Q = M
    while Q > 1:          # inverse undo
        P = Q - np.uint64(1)
        for i in range(n):
            hit = (X[i] & Q) != 0
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = (X[0] ^ X[i]) & P
            t = np.where(hit, np.uint64(0), t)
            X[0] ^= t
            X[i] ^= t
        Q >>= np.uint64(1)
    for i in range(1, n):  # Gray encode
        X[i] ^= X[i - 1]
    t = np.zeros(X.shape[1], np.uint64)
    Q = M
    while Q > 1:
        t = np.where((X[n - 1] & Q) != 0, t ^ (Q - np.uint64(1)), t)
        Q >>= np.uint64(1)
    for i in range(n):
        X[i] ^= t
    key = np.zeros(X.shape[1], np.uint64)
    for b in range(bits - 1, -1, -1):
        for i in range(n):
            key = (key << np.uint64(1)) | ((X[i] >> np.uint64(b)) & np.uint64(1))
    return key


def relabel(adj, order):
    n = adj.shape[0] // 4
    rank = np.empty(n, np.int64)
    rank[order] = np.arange(n)
    nb = adj[:, 1].reshape(n, 4)[order]
    return rank[nb]          # [n, 4]: neighbour ids of the cells in the new order


def stack_distances(req):
    """stack distance (distinct rows since the last use; -1 = first use) of every request, O(N log N) with a Fenwick tree over request times"""
    N = len(req)
    tree = [0] * (N + 1)
    last = {}
    out = np.empty(N, np.int64)

    def add(i, v):
        i += 1
        while i <= N:
            tree[i] += v
            i += i & -i

    def prefix(i):      # sum of [0, i)
        s = 0
        while i > 0:
            s += tree[i]
            i -= i & -i
        return s
    live = 0
    for t, r in enumerate(req):
        p = last.get(r)
        if p is None:
            out[t] = -1
        else:
            out[t] = live - prefix(p + 1)        # distinct rows whose last use is after p
            add(p, -1)
            live -= 1
        add(t, 1)
        live += 1
        last[r] = t
    return out


def main():
    points = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    adj, cent, _ = delaunay_tet_graph(points, 0)
    n = adj.shape[0] // 4
    _, cent_m, order_m = loader_cell_order(adj, cent)
    orders = {"loader (Morton)": order_m, "Hilbert": np.argsort(hilbert_keys(cent), kind="stable"), "generator": np.arange(n)}
    caps = [512, 1024, 2048, 3072, 4096, 8192, 16384, 65536]
    print("scene: %d points, %d cells; one XCD's eighth = %d cells, 4 neighbour-row requests per cell; cache capacities in ROWS (4096 rows = the 4 MB L2 at 1 KB rows, "
          "C = 256; 8192 at 512 B rows, C = 128)" % (points, n, n // 8))
    print("%-18s %s   first-use   median |src - dst|   far (> 2048 rows)" % ("order", "  ".join("miss@%-6d" % c for c in caps)))
    for name, order in orders.items():
        nb = relabel(adj, order)
        lo, hi = 3 * (n // 8), 4 * (n // 8)            # an inner eighth
        req = nb[lo:hi].reshape(-1)
        sd = stack_distances(req.tolist())
        first = float((sd < 0).mean())
        d = np.abs(nb - np.arange(n)[:, None])
        row = "  ".join("%10.3f" % float(((sd < 0) | (sd > c)).mean()) for c in caps)
        print("%-18s %s   %.3f       %6d              %.3f" % (name, row, first, int(np.median(d)), float((d > 2048).mean())))
    print("reads per unique row at capacity R = 4 x miss@R (a row is requested by its 4 neighbours; 1.0 = every row fetched once)")


if __name__ == "__main__":
    main()
