"""npz ingest producing the kernels' layouts directly (reference processing/data.py).

``dataLoader(clf).run(d)`` keeps the reference's interface (:81-112): it reads
``<path>/<gtfile>_{labels,cgeom,cbvf,cbff,adjacencies,fgeom,fbvf,fbff}.npz``, selects feature columns exactly as
``readNodeData_bin`` (:194-284) and ``readEdgeData_bin`` (:353-414) do (same order, same 'last'-column drops, the
un-scaled ``reg_<cell_type>`` copy in column 0), and leaves ``features`` [N, 1+F] fp32, ``edge_features`` [E, F_e]
fp32, ``edge_lists`` int64 [2,E], ``gt``, ``infinite`` ON THE GPU.  Column selection is host-side bookkeeping on
numpy arrays (no pandas); the per-scene StandardScaler (:471-506) runs on the device in fp64
(dgnn_standardize_f64).  Only scaling 's' (all shipped configs) is implemented; others exit like the reference.

Cell order (round 4): the reference builds ``edge_index`` straight from the file (:434-438), i.e. in CGAL's insertion order -- the 4
neighbours of a cell are tens of thousands of rows apart and every neighbour gather of the conv layers misses L2.  ``run`` relabels the cells
once per scene (``processing/reorder.py``: Morton order of the cell centroids from ``<scene>_3dt.npz``, breadth-first order of the adjacency when
the scene has no coordinates), permutes ``features / edge_features / edge_lists / gt / infinite`` consistently and keeps the permutation as
``cell_order``; ``exportScore`` and ``generate_mesh.generate`` put per-cell results back in file order.  ``clf.temp.cell_order`` = "auto" (default)
| "morton" | "bfs" | "none" (or DGNN_CELL_ORDER in the environment).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from .._lib import check, lib, ptr, stream_ptr
from .. import ops
from . import reorder


def standardize(cols64: np.ndarray, c_first: int, device) -> torch.Tensor:
    """[N,C] float64 (host) -> standardised fp32 [N,C] on `device`; columns < c_first are only cast."""
    with torch.cuda.device(device):
        x = torch.from_numpy(np.ascontiguousarray(cols64)).to(device)
        n, c = x.shape
        out = torch.empty((n, c), dtype=torch.float32, device=device)
        scratch = torch.empty(int(lib().dgnn_standardize_scratch_doubles(c)), dtype=torch.float64, device=device)
        check(lib().dgnn_standardize_f64(ptr(x), c, n, c, c_first, ptr(out), c, ptr(scratch), stream_ptr()), "dgnn_standardize_f64")
    return out


class dataLoader:
    STATS = ("count", "min", "max", "sum")

    def __init__(self, clf, verbosity=1):
        self.clf = clf
        self.verbosity = verbosity
        self.read_edge_features = clf.model.edge_convs
        self.n_nodes = 0

    # -- column selection (host) ---------------------------------------------------------------------------------
    @staticmethod
    def _stat(key):
        return key.rsplit("_", 1)[-1]            # count | min | max | sum for the cb_*/fb_* columns

    @staticmethod
    def _npz_drop_quirk(wanted, what):
        # The reference drops the per-statistic columns of these groups with `temp.drop(...)` on the NpzFile
        # (data.py:240-251, 365-373, 379-387), which has no such method: a config that leaves a statistic out
        # fails there.  Same behaviour, explicit message.
        missing = [s for s in dataLoader.STATS if s not in wanted]
        if missing:
            raise AttributeError("'NpzFile' object has no attribute 'drop' (reference %s columns cannot omit %s)" % (what, missing))

    def _node_columns(self, base):
        f = self.clf.features.node_features
        names, cols = [], []
        temp = np.load(base + "_cgeom.npz")
        self.mean_edge = (temp["longest_edge"].sum() + temp["shortest_edge"].sum()) / (2 * len(temp["longest_edge"]))
        if "shape" in f:
            for k in temp.files:
                names.append(k); cols.append(temp[k])
        drop_last = "last" not in f
        if "vertex" in f:
            t = np.load(base + "_cbvf.npz")
            for k in t.files:
                if self._stat(k) in f and not (drop_last and "_last_" in k):       # :226-233, :254-263
                    names.append(k); cols.append(t[k])
        if "facet" in f:
            t = np.load(base + "_cbff.npz")
            self._npz_drop_quirk(f, "cb_facet")
            for k in t.files:
                if not (drop_last and "_last_" in k):                              # :264-272
                    names.append(k); cols.append(t[k])
        reg = self.clf.regularization.cell_type
        if reg:                                                                    # :275-276
            cols.insert(0, cols[names.index(reg)]); names.insert(0, "reg_" + reg)
        return names, np.stack(cols, axis=1).astype(np.float64)

    def _edge_columns(self, base):
        f = self.clf.features.edge_features
        names, cols = [], []
        if "shape" in f:
            t = np.load(base + "_fgeom.npz")
            for k in t.files:
                names.append(k); cols.append(t[k])
        drop_last = "last" not in f
        for tag, fn in (("vertex", "_fbvf.npz"), ("facet", "_fbff.npz")):
            if tag in f:
                t = np.load(base + fn)
                self._npz_drop_quirk(f, "fb_" + tag)
                for k in t.files:
                    if not (drop_last and "_last_" in k):                          # :391-410
                        names.append(k); cols.append(t[k])
        reg = self.clf.regularization.edge_type
        if reg:                                                                    # :413-414
            cols.insert(0, cols[names.index(reg)]); names.insert(0, "reg_" + reg)
        return names, np.stack(cols, axis=1).astype(np.float64)

    # -- the reference's entry point -------------------------------------------------------------------------------
    def run(self, d):
        dev = self.clf.temp.device
        self.path, self.filename, self.gtfile = d["path"], d["filename"], d["gtfile"]
        self.category, self.id, self.scan_conf, self.ioufile = d["category"], d["id"], d["scan_conf"], d["ioufile"]
        base = self.basefilename = os.path.join(self.path, self.gtfile)
        lab = np.load(base + "_labels.npz")
        if self.clf.inference.has_label:
            self.gt = torch.from_numpy(np.stack([lab["inside_perc"], lab["outside_perc"]], 1).astype(np.float32)).to(dev)
        else:
            self.gt = torch.zeros(lab["infinite"].shape, device=dev)
        self.infinite = torch.from_numpy(lab["infinite"]).to(dev).bool()
        self.node_feature_names, nodes = self._node_columns(base)
        adj = np.load(base + "_adjacencies.npz")["adjacencies"]
        # the reference's own layout (:437-438): the [E,2] array on the device, handed on as its transposed view (strides (1,2)) -- read in place by the
        # plan builder, whose four-lanes-per-cell pass takes exactly this view
        self.edge_lists = torch.from_numpy(np.ascontiguousarray(adj.astype(np.int64))).to(dev).t()
        scaling = self.clf.features.scaling
        if scaling != "s" or self.clf.features.node_normalization_feature is not None \
                or self.clf.features.edge_normalization_feature is not None:
            # every shipped config (configs/*.yaml:2-7) uses scaling 's' and no normalisation feature
            print("scaling {!r} / normalization features are not supported by dgnn_amd; use scaling: s".format(scaling))
            sys.exit(1)
        c_first = 1 if self.clf.regularization.cell_type else 0
        self.features = standardize(nodes, c_first, dev)
        if self.read_edge_features:
            self.edge_feature_names, edges = self._edge_columns(base)
            assert adj.shape[0] == edges.shape[0]
            self.edge_features = standardize(edges, 1 if self.clf.regularization.edge_type else 0, dev)
        else:
            self.edge_features = torch.empty(1, 1, dtype=torch.float32, device=dev)
        self.n_nodes += self.features.size(0)
        self.cell_order = None
        kind = str(getattr(self.clf.temp, "cell_order", None) or os.environ.get("DGNN_CELL_ORDER", "auto")).lower()
        if kind not in ("none", "0", "false", "off"):
            self._reorder_cells(base, kind)
        reorder.register_scene_order(self, self.cell_order)     # by path + gtfile: survives every copy of the loader's tensors (reorder.find_cell_order)

    def _reorder_cells(self, base, kind):
        """relabels the scene (module docstring); every per-cell / per-edge tensor of the loader moves with its cell"""
        n = self.features.size(0)
        ei = self.edge_lists
        co = reorder.scene_order(ei, n, infinite=self.infinite, mfile=base + "_3dt.npz", kind=kind)
        if co is None:
            return
        ei_new, edge_rows = reorder.reorder_edges(ei, co.order, co.rank)
        self.features = ops.gather_rows(self.features, co.order)
        if self.read_edge_features:
            self.edge_features = ops.gather_rows(self.edge_features, edge_rows)
        self.edge_lists = ei_new
        self.gt = co.to_rows(self.gt)
        self.infinite = co.to_rows(self.infinite)
        self.cell_order = co
        for t in (self.features, self.gt, self.infinite):      # an unmodified run.py:prepareSample copies these tensors, not the loader's fields
            setattr(t, reorder.RANK_TAG, co)

    def exportScore(self, prediction):
        """reference :521-535 -- the scores of the cells in FILE order (the order every consumer of <out>/prediction/<scene>.npz knows)"""
        outpath = os.path.join(self.clf.paths.out, "prediction")
        os.makedirs(outpath, exist_ok=True)
        if self.verbosity:
            print("Export predictions to: ", outpath)
        prediction = reorder.restore_cell_order(torch.as_tensor(prediction), self).detach().cpu()
        with open(os.path.join(outpath, self.filename + ".npz"), "wb") as f:
            np.savez(f, number_of_cells=int(len(prediction)), sigmoid=prediction.sigmoid().numpy(), logits=prediction.numpy(),
                     softmax=prediction.softmax(dim=-1).numpy())

    def getInfo(self):
        """Sets clf.temp.num_{node,edge}_features as the reference does (:39-48)."""
        self.clf.temp.num_node_features = self.features.size(1) - bool(self.clf.regularization.cell_type)
        self.clf.temp.num_edge_features = (self.edge_features.size(1) - bool(self.clf.regularization.edge_type)) \
            if self.read_edge_features else None
        return self.n_nodes
