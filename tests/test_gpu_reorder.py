"""Ingest-time cell locality order (csrc/reorder.hip, dgnn_amd/processing/reorder.py): index work, so every
piece is held BIT-EXACT to a numpy restatement written here (test infrastructure), and the relabelled scene must give the logits of
the scene in file order (the same graph, sums in another order)."""
import collections
import os

import numpy as np
import pytest
import torch

from dgnn_amd.config import Config
from helpers import gold
from test_gpu_parity import DEV, hip_static

pytestmark = pytest.mark.gpu


# ---- numpy restatements ------------------------------------------------------------------------------------------------------
def np_morton_keys(cent):
    c = np.asarray(cent, np.float32)
    lo = np.nanmin(c, axis=0).astype(np.float32) if len(c) else np.zeros(3, np.float32)
    hi = np.nanmax(c, axis=0).astype(np.float32) if len(c) else np.zeros(3, np.float32)
    ext = (hi - lo).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(ext > 0, np.float32(65535.0) / ext, np.float32(0)).astype(np.float32)
    q = ((c - lo).astype(np.float32) * inv).astype(np.float32)
    q = np.where(np.isnan(q), np.float32(65535), np.minimum(np.maximum(q, np.float32(0)), np.float32(65535)))
    qi = q.astype(np.uint64)
    keys = np.zeros(len(c), np.uint64)
    for a in range(3):
        for b in range(16):
            keys |= ((qi[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return keys


def np_bfs_order(dst4):
    """serial queue BFS over rows of 4 neighbours, components started at their lowest cell id"""
    n = dst4.shape[0]
    seen = np.zeros(n, bool)
    order = []
    for s in range(n):
        if seen[s]:
            continue
        seen[s] = True
        q = collections.deque([s])
        while q:
            u = q.popleft()
            order.append(u)
            for v in dst4[u]:
                if not seen[v]:
                    seen[v] = True
                    q.append(v)
    return np.asarray(order, np.int64)


def small_delaunay(points, seed=0):
    from dgnn_amd.synthetic import delaunay_tet_graph
    adj, cent, nf = delaunay_tet_graph(points, seed=seed)
    return adj, cent, nf


# ---- pieces --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 2047, 2048, 2049, 70001])
def test_morton_order_is_the_stable_sort_of_the_keys(n):
    from dgnn_amd.processing.reorder import cell_order_morton
    rng = np.random.default_rng(n)
    cent = rng.standard_normal((n, 3)).astype(np.float32) * np.array([1.0, 50.0, 1e-3], np.float32)
    if n > 100:
        cent[rng.integers(0, n, n // 10)] = cent[rng.integers(0, n, n // 10)]     # exact duplicates: stability decides
        cent[5] = np.nan                                                          # a NaN coordinate sorts to the far corner, bbox ignores it
        cent[n // 2:n // 2 + 40] = cent[3]
    order, rank = cell_order_morton(torch.from_numpy(cent).to(DEV))
    want = np.argsort(np_morton_keys(cent), kind="stable")
    assert np.array_equal(order.cpu().numpy(), want)
    assert np.array_equal(rank.cpu().numpy()[want], np.arange(n))


def test_morton_order_of_identical_points_is_the_identity():
    from dgnn_amd.processing.reorder import cell_order_morton
    order, _ = cell_order_morton(torch.ones(5000, 3, device=DEV))
    assert np.array_equal(order.cpu().numpy(), np.arange(5000))


def test_centroids_from_3dt_vs_numpy():
    from dgnn_amd.processing.reorder import centroids_from_3dt
    rng = np.random.default_rng(1)
    n, nv = 6000, 900
    infinite = (rng.random(n) < 0.2).astype(np.int32)
    nf = int((infinite == 0).sum())
    verts = rng.standard_normal((nv, 3))
    tets = rng.integers(0, nv, (nf, 4)).astype(np.int64)
    dst = rng.integers(0, n, (n, 4))
    adj = np.stack([np.repeat(np.arange(n), 4), dst.reshape(-1)], 1).astype(np.int64)
    ei = torch.from_numpy(adj).to(DEV).t()
    got = centroids_from_3dt(torch.from_numpy(verts), torch.from_numpy(tets), torch.from_numpy(infinite), ei).cpu().numpy()
    v32 = verts.astype(np.float32)
    fin_rank = np.cumsum(infinite == 0) - (infinite == 0)
    want = np.zeros((n, 3), np.float32)
    for i in range(n):
        cell = i if infinite[i] == 0 else next((d for d in dst[i] if infinite[d] == 0), None)
        if cell is None:
            continue
        p = v32[tets[fin_rank[cell]]]
        want[i] = np.float32(0.25) * (((p[0] + p[1]) + p[2]) + p[3])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("points", [40, 900])
def test_bfs_order_is_the_serial_queue_order(points):
    from dgnn_amd.processing.reorder import cell_order_bfs
    adj, _, _ = small_delaunay(points, seed=3)
    n = adj.shape[0] // 4
    perm = np.random.default_rng(0).permutation(n)          # CGAL-like labels
    dst4 = perm[adj[:, 1].astype(np.int64)].reshape(n, 4)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    dst4 = dst4[inv]                                         # row of new cell j = neighbours of old cell inv[j]
    pairs = np.stack([np.repeat(np.arange(n), 4), dst4.reshape(-1)], 1).astype(np.int64)
    order, rank = cell_order_bfs(torch.from_numpy(pairs).to(DEV).t(), n)
    want = np_bfs_order(dst4)
    assert np.array_equal(order.cpu().numpy(), want)
    assert np.array_equal(rank.cpu().numpy()[want], np.arange(n))


def test_bfs_order_walks_every_component():
    from dgnn_amd.processing.reorder import cell_order_bfs
    a1, _, _ = small_delaunay(30, seed=1)
    a2, _, _ = small_delaunay(45, seed=2)
    n1, n2 = a1.shape[0] // 4, a2.shape[0] // 4
    n = n1 + n2 + 1
    dst4 = np.concatenate([a1[:, 1].astype(np.int64).reshape(n1, 4), a2[:, 1].astype(np.int64).reshape(n2, 4) + n1, np.full((1, 4), n - 1)])   # + a cell that only knows itself
    pairs = np.stack([np.repeat(np.arange(n), 4), dst4.reshape(-1)], 1).astype(np.int64)
    order, _ = cell_order_bfs(torch.from_numpy(pairs).to(DEV).t(), n)
    assert np.array_equal(order.cpu().numpy(), np_bfs_order(dst4))


def test_reordered_edges_vs_numpy_and_a_foreign_layout_is_reported():
    from dgnn_amd._lib import DgnnError
    from dgnn_amd.processing.reorder import reorder_edges
    adj, _, _ = small_delaunay(300, seed=5)
    n = adj.shape[0] // 4
    order = np.random.default_rng(2).permutation(n).astype(np.int32)
    rank = np.empty(n, np.int32)
    rank[order] = np.arange(n, dtype=np.int32)
    for ei in (torch.from_numpy(adj.astype(np.int64)).to(DEV).t(), torch.from_numpy(np.ascontiguousarray(adj.T.astype(np.int64))).to(DEV)):
        new, rows = reorder_edges(ei, torch.from_numpy(order).to(DEV), torch.from_numpy(rank).to(DEV))
        assert new.stride() == (1, 2)
        want_rows = (4 * order.astype(np.int64)[:, None] + np.arange(4)).reshape(-1)
        assert np.array_equal(rows.cpu().numpy(), want_rows)
        assert np.array_equal(new[0].cpu().numpy(), np.repeat(np.arange(n), 4))
        assert np.array_equal(new[1].cpu().numpy(), rank[adj[want_rows, 1]])
    bad = adj.astype(np.int64).copy()
    bad[[4, 9]] = bad[[9, 4]]          # rows 4 and 9 swapped: row 4 no longer leaves cell 1
    with pytest.raises(DgnnError):
        reorder_edges(torch.from_numpy(bad).to(DEV).t(), torch.from_numpy(order).to(DEV), torch.from_numpy(rank).to(DEV))
        torch.cuda.synchronize()
        from dgnn_amd._lib import check
        check(0, "poll", poll=True)


# ---- the relabelled scene gives the scene's logits ---------------------------------------------------------------------------------
def _scrambled_scene(points, seed):
    """a Delaunay scene whose cells carry random labels (what CGAL's insertion order looks like to the gathers), with its centroids"""
    import bench
    adj, cent, x, ea = bench.make_scene(points, seed)
    n = x.shape[0]
    perm = np.random.default_rng(seed + 1).permutation(n)              # new id of old cell
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    dst4 = perm[adj[:, 1].astype(np.int64)].reshape(n, 4)[inv]
    pairs = np.stack([np.repeat(np.arange(n), 4), dst4.reshape(-1)], 1).astype(np.int64)
    rows = (4 * inv[:, None] + np.arange(4)).reshape(-1)
    return dict(n=n, x=x[inv], ea=ea[rows], pairs=pairs, cent=cent[inv], x0=x, ea0=ea, adj0=adj, perm=perm)


@pytest.mark.parametrize("kind", ["morton", "bfs"])
def test_reordered_scene_gives_the_same_logits_and_a_local_graph(kind):
    from dgnn_amd import ops
    from dgnn_amd.processing.reorder import reorder_edges, scene_order
    s = _scrambled_scene(6000, 4)
    n = s["n"]
    net = hip_static()
    ei = torch.from_numpy(s["pairs"]).to(DEV).t()
    x, ea = s["x"].to(DEV), s["ea"].to(DEV)
    base = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=ei))
    co = scene_order(ei, n, centroids=torch.from_numpy(s["cent"]).to(DEV) if kind == "morton" else None, kind=kind)
    assert co.kind == kind
    ei2, rows = reorder_edges(ei, co.order, co.rank)
    x2, ea2 = ops.gather_rows(x, co.order), ops.gather_rows(ea, rows)
    got = net.inference_layer(Config(x=x2, edge_attr=ea2, edge_index=ei2))
    assert (co.to_file(got) - base).abs().max().item() <= 2e-5
    # locality: the median |src - dst| row distance collapses (random labels: ~n/3)
    d_old = (ei[0] - ei[1]).abs().float().median().item()
    d_new = (ei2[0] - ei2[1]).abs().float().median().item()
    # (a breadth-first shell of a 40k-cell ball holds ~3k cells: neighbours sit about one shell apart; Morton blocks are tighter)
    assert d_old > n / 6 and d_new < (n / 40 if kind == "morton" else n / 16), (d_old, d_new)


@pytest.mark.parametrize("kind", ["morton", "bfs"])
def test_ignatius_full_scene_reordered_vs_reference_logits(kind):
    """the whole real scene (67 017 cells in CGAL's order: median |src - dst| 753 rows, a fifth of the edges further than 16k rows) through the
    loader's orders -- Morton from the scene's own `_3dt` geometry (tests/golden/ignatius_3dt.npz), breadth-first from the adjacency alone --:
    logits restored to file order against the REFERENCE's, and what the order does to the neighbour distances"""
    from dgnn_amd import ops
    from dgnn_amd.processing.reorder import centroids_from_3dt, reorder_edges, scene_order
    from test_gpu_scale import logit_check
    g = gold("static_f4_ignatius_full.npz")
    n = g["x"].shape[0]
    fg = np.random.default_rng(int(g["fgeom_seed"])).standard_normal((4 * n, 4)).astype(np.float32)
    ea = torch.from_numpy(np.concatenate([fg, g["edge_attr16"]], axis=1)).to(DEV)
    pairs = np.stack([np.repeat(np.arange(n, dtype=np.int64), 4), g["adj_dst"].astype(np.int64)], 1)
    ei = torch.from_numpy(pairs).to(DEV).t()
    x = torch.from_numpy(np.ascontiguousarray(g["x"])).to(DEV)
    cent = None
    if kind == "morton":
        m = gold("ignatius_3dt.npz")
        cent = centroids_from_3dt(torch.from_numpy(m["vertices"]), torch.from_numpy(m["tetrahedra"]), torch.from_numpy(m["infinite"]), ei)
        fin = m["infinite"] == 0                       # numpy restatement on the real geometry (fp32 like the kernel)
        want = np.zeros((n, 3), np.float32)
        p = m["vertices"].astype(np.float32)[m["tetrahedra"]]
        want[fin] = np.float32(0.25) * (((p[:, 0] + p[:, 1]) + p[:, 2]) + p[:, 3])
        dst4 = g["adj_dst"].astype(np.int64).reshape(n, 4)
        for i in np.nonzero(~fin)[0]:
            nb = [d for d in dst4[i] if fin[d]]
            if nb:
                want[i] = want[nb[0]]
        assert np.array_equal(cent.cpu().numpy(), want)
    co = scene_order(ei, n, centroids=cent, kind=kind)
    assert co.kind == kind
    ei2, rows = reorder_edges(ei, co.order, co.rank)
    net = hip_static()
    got = net.inference_layer(Config(x=ops.gather_rows(x, co.order), edge_attr=ops.gather_rows(ea, rows), edge_index=ei2))
    err = logit_check(co.to_file(got).cpu().numpy(), g["logits"], g["logits64"])
    d_old = (ei[0] - ei[1]).abs().float()
    d_new = (ei2[0] - ei2[1]).abs().float()
    far = lambda d: (d > 16384).float().mean().item()
    print("Ignatius reordered (%s): max|dlogit| %.3e; |src-dst| median %d -> %d rows, mean %.0f -> %.0f, further than 16k rows %.3f -> %.3f"
          % (kind, err, d_old.median().item(), d_new.median().item(), d_old.mean().item(), d_new.mean().item(), far(d_old), far(d_new)))
    assert far(d_old) > 0.15                                           # CGAL's order: a heavy tail of far neighbours
    if kind == "morton":
        assert d_new.median().item() <= 32 and d_new.mean().item() < d_old.mean().item() / 5 and far(d_new) < 0.03
    else:                                                              # breadth-first: every neighbour within a couple of shells, no tail at all
        assert d_new.max().item() < n / 4 and d_new.mean().item() < d_old.mean().item() / 2 and far(d_new) == 0.0


def test_generate_takes_a_reordered_scene(tmp_path):
    """generate(data, prediction, clf) on a relabelled scene: `_3dt.npz` speaks file order -- same mesh as the reference's run in file order"""
    from dgnn_amd.processing.generate_mesh import generate
    from dgnn_amd.processing.reorder import RANK_TAG, CellOrder
    g = gold("genmesh_f4_small.npz")
    os.makedirs(os.path.join(str(tmp_path), "gt"))
    np.savez(os.path.join(str(tmp_path), "gt", "0_3dt.npz"), vertices=g["vertices"], tetrahedra=g["tetrahedra"], facets=g["facets"], nfacets=g["nfacets"])
    n = g["infinite"].shape[0]
    order = torch.from_numpy(np.random.default_rng(0).permutation(n).astype(np.int32)).to(DEV)
    rank = torch.empty_like(order)
    rank[order.long()] = torch.arange(n, dtype=torch.int32, device=DEV)
    co = CellOrder(order, rank, "test")
    inf_rows = co.to_rows(torch.from_numpy(g["infinite"]).to(DEV))
    pred_rows = co.to_rows(torch.from_numpy(g["prediction"]).to(DEV))
    clf = Config(temp=Config(graph_cut=0, fix_orientation=0, metrics=[], device=DEV), graph_cut=Config(unary_weight=10.0, binary_weight=1.0, binary_type=0))
    # (a) the dataset object carries the order; (b) only the loader's tensor does (an unmodified run.py:prepareSample)
    setattr(inf_rows, RANK_TAG, co)
    for data in (Config(path=str(tmp_path), gtfile="gt/0", filename="0", id="", category="", infinite=inf_rows.clone(), cell_order=co),
                 Config(path=str(tmp_path), gtfile="gt/0", filename="0", id="", category="", infinite=inf_rows)):
        mesh, _ = generate(data, pred_rows, clf)
        assert np.array_equal(np.asarray(mesh.faces), g["faces"]) and np.array_equal(np.asarray(mesh.vertices), g["vertices_out"])
    # (c) ADVICE r4: every tensor was COPIED (.clone() / .cpu() / cat drop the tag) -- the order is found by the scene's path + gtfile in the
    # registry the loader fills; a permutation of another length raises instead of scrambling the mesh
    from dgnn_amd.processing import reorder as R
    key_obj = Config(path=str(tmp_path), gtfile="gt/0", infinite=inf_rows.clone())      # what dataLoader.run registers: the loader after relabelling
    R.register_scene_order(key_obj, co)
    try:
        data = Config(path=str(tmp_path), gtfile="gt/0", filename="0", id="", category="", infinite=inf_rows.clone().cpu())
        assert getattr(data.infinite, RANK_TAG, None) is None
        mesh, _ = generate(data, pred_rows.clone(), clf)
        assert np.array_equal(np.asarray(mesh.faces), g["faces"]) and np.array_equal(np.asarray(mesh.vertices), g["vertices_out"])
        with pytest.raises(RuntimeError, match="permutation"):
            R.restore_cell_order(pred_rows[:-1], data)
        # an object with the same files whose rows are in FILE order (built straight from the npz, a cached dataset, a loader with
        # cell_order off that did not register) is NOT permuted: its `infinite` flags are not the relabelled scene's
        in_file_order = Config(path=str(tmp_path), gtfile="gt/0", filename="0", id="", category="", infinite=torch.from_numpy(g["infinite"]).clone())
        assert R.find_cell_order(in_file_order)[0] is None
        mesh, _ = generate(in_file_order, torch.from_numpy(g["prediction"]).to(DEV), clf)
        assert np.array_equal(np.asarray(mesh.faces), g["faces"])
        # ... nor one that carries no per-cell flags at all; and the registry is bounded
        assert R.find_cell_order(Config(path=str(tmp_path), gtfile="gt/0"))[0] is None
        for i in range(R._SCENE_ORDERS_MAX + 5):
            R.register_scene_order(Config(path=str(tmp_path), gtfile="gt/other%d" % i, infinite=inf_rows), co)
        assert len(R._SCENE_ORDERS) == R._SCENE_ORDERS_MAX
        R.register_scene_order(key_obj, co)
        R.register_scene_order(key_obj, None)         # the same scene loaded again WITHOUT relabelling: the stale order is gone
        assert R.find_cell_order(data) == (None, True)
    finally:
        R._SCENE_ORDERS.clear()
