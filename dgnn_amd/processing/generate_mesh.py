"""Logits -> labels -> interface facets on the GPU (reference processing/generate_mesh.py:61-107).

``extract_interface(prediction, infinite, nfacets)`` reproduces what ``generate`` does between the network output
and the trimesh call: labels of the finite cells (:75), the infinite cell appended as OUTSIDE (:93-99) and the list
of facets whose two cells carry different labels (:101-105) -- two Python loops over all facets in the reference,
three small kernels here.  The optional integer alpha-expansion graph cut (:15-58, third-party ``gco``) and the
``trimesh`` mesh object are CPU-side third-party steps and stay outside; ``labels`` can be replaced by the
graph-cut labels before ``interface_from_labels``.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .._lib import check, lib, on_device_of, ptr, stream_ptr


@on_device_of
def _compact(values, keep, invert):
    n = keep.numel()
    out = torch.empty(max(n, 1), dtype=torch.int32, device=keep.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=keep.device)
    scratch = torch.empty(int(lib().dgnn_compact_scratch_elems(n)), dtype=torch.int32, device=keep.device)
    check(lib().dgnn_compact_i32(ptr(values), ptr(keep), int(invert), n, ptr(out), ptr(cnt), ptr(scratch), stream_ptr()),
          "dgnn_compact_i32")
    return out[:int(cnt.item())]


@on_device_of
def labels_of_finite_cells(prediction: torch.Tensor, infinite: torch.Tensor) -> torch.Tensor:
    """int32 labels (0 inside / 1 outside) of the cells with infinite == 0, in cell order (reference :75)."""
    if not prediction.is_cuda:
        raise RuntimeError("prediction must be on the GPU")
    prediction = prediction.contiguous()
    n = prediction.size(0)
    labels = torch.empty(n, dtype=torch.int32, device=prediction.device)
    check(lib().dgnn_argmax_rows(ptr(prediction), prediction.size(1), n, prediction.size(1), ptr(labels), stream_ptr()), "dgnn_argmax_rows")
    return _compact(labels, infinite.to(prediction.device, torch.int32).contiguous(), invert=True)


@on_device_of
def interface_from_labels(labels_finite: torch.Tensor, nfacets: torch.Tensor) -> torch.Tensor:
    """Indices (int32, ascending) of the facets whose two cells differ; cell -1 is the outside cell (:93-105)."""
    nfacets = nfacets.to(labels_finite.device, torch.int32).contiguous()
    f = nfacets.size(0)
    flags = torch.empty(max(f, 1), dtype=torch.int32, device=labels_finite.device)[:f]
    check(lib().dgnn_interface_flags(ptr(nfacets), ptr(labels_finite.contiguous()), f, ptr(flags), stream_ptr()), "dgnn_interface_flags")
    return _compact(None, flags, invert=False)


def extract_interface(prediction: torch.Tensor, infinite: torch.Tensor, nfacets: torch.Tensor):
    """-> (labels_finite int32 [Nf], interface facet ids int32 [n_interface])"""
    labels = labels_of_finite_cells(prediction, infinite)
    return labels, interface_from_labels(labels, nfacets)


# ---- reference entry point (processing/generate_mesh.py:61-165) ------------------------------------------------------------
class InterfaceMesh:
    """What `generate` returns as the mesh when trimesh is not installed: the interface triangles as plain arrays, with the
    one method the callers use (`export`, run.py:191 / learning/runModel.py:360).  With trimesh present `generate` returns a
    trimesh.Trimesh built exactly as the reference does (:107-111)."""

    def __init__(self, vertices: np.ndarray, faces: np.ndarray):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)

    def export(self, path: str):
        """binary little-endian PLY (vertices double x/y/z, faces as uchar-count int32 lists)"""
        v, f = self.vertices, self.faces.astype(np.int32)
        with open(path, "wb") as fh:
            fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\n"
                      "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(v), len(f))).encode())
            fh.write(np.ascontiguousarray(v, dtype="<f8").tobytes())
            rec = np.empty(len(f), dtype=[("n", "u1"), ("i", "<i4", (3,))])
            rec["n"] = 3
            rec["i"] = f
            fh.write(rec.tobytes())
        return path


def graph_cut(labels, prediction, edges, clf):
    """The reference's alpha-expansion smoothing (:15-58) with its INTEGER costs: unary = round(swapped logits *
    graph_cut.unary_weight), Potts smoothness * binary_weight.  The solver is the third-party `gco` wrapper (un-vendored,
    environment.yml:115) and stays on the CPU; ImportError propagates to `generate`, which falls back to the raw labels
    exactly as the reference's bare `except` does (:88-91)."""
    import gco  # noqa: F401  (gco-wrapper 3.0.8)

    dtype = np.int64
    gc = gco.GCO()
    gc.create_general_graph(int(edges.max()) + 1, 2, energy_is_float=False)
    pred = np.asarray(prediction, dtype=np.float64)[:, [1, 0]]
    gc.set_data_cost(np.array((pred * clf.graph_cut.unary_weight).round(), dtype=dtype))
    gc.set_smooth_cost((1 - np.eye(2)).astype(dtype))
    gc.set_all_neighbors(edges[:, 0], edges[:, 1], np.ones(edges.shape[0], dtype=dtype) * clf.graph_cut.binary_weight)
    for i, l in enumerate(labels):
        gc.init_label_at_site(i, l)
    gc.expansion()
    return gc.get_labels()


def generate(data, prediction, clf):
    """Same signature and return value as the reference's processing/generate_mesh.py:61 ``generate(data, prediction, clf)``
    -> ``(mesh, eval_dict)``; what runs where:

    * labels of the finite cells (``log_softmax(prediction[infinite == 0]).argmax(1)``, :75) and the interface facets
      (``labels[f0] != labels[f1]`` over all facets with the infinite cell = outside, :93-105 -- two nested Python loops in
      the reference) run on the GPU (dgnn_argmax_rows / dgnn_interface_flags / dgnn_compact_i32); integer results, identical;
    * the optional graph cut (``clf.temp.graph_cut``) is the reference's CPU solver when `gco` imports, otherwise the raw
      labels are kept with the reference's warning;
    * the mesh object is a trimesh.Trimesh (``process=True``, optional fix_normals) when trimesh imports, else an
      InterfaceMesh with the same vertices / faces and an ``export``; the evaluation metrics (watertight / iou / chamfer,
      :115-163) need trimesh + utils/libmesh and are computed only when those import -- otherwise eval_dict stays empty.
    """
    dev = prediction.device if prediction.is_cuda else torch.device(getattr(clf.temp, "device", "cuda:0"))
    pred_dev = prediction.to(dev, torch.float32)
    infinite = torch.as_tensor(data.infinite)
    # a scene the loader relabelled (processing/reorder.py): `_3dt.npz` knows the cells in FILE order -- logits and the infinite flags go back to it
    from .reorder import restore_cell_order
    # (found on the object, on the loader's tagged tensors, or -- when those were copied -- in the registry of loaded scenes by path + gtfile)
    pred_dev, prediction = restore_cell_order(pred_dev, data), restore_cell_order(prediction, data)
    infinite = restore_cell_order(infinite, data)
    mfile = os.path.join(data.path, data.gtfile + "_3dt.npz")
    mdata = np.load(mfile)
    nfacets = np.ascontiguousarray(mdata["nfacets"]).astype(np.int32)
    labels_dev = labels_of_finite_cells(pred_dev, infinite)
    assert labels_dev.numel() == len(mdata["tetrahedra"])
    if getattr(clf.temp, "graph_cut", None):
        mask = (nfacets >= 0).all(axis=1)
        try:
            finite = (infinite == 0).to(prediction.device)
            lab = graph_cut(labels_dev.cpu().numpy(), prediction[finite].detach().cpu().numpy(), nfacets[mask], clf)
            labels_dev = torch.as_tensor(np.asarray(lab), dtype=torch.int32, device=dev)
        except Exception:  # noqa: BLE001  (the reference: bare except, :88-91)
            print("WARNING: Graph cut for {} didn't work. Using raw predictions for mesh generation.".format(data.filename))
    interfaces = interface_from_labels(labels_dev, torch.from_numpy(nfacets)).cpu().numpy()
    faces = mdata["facets"][interfaces]
    eval_dict = dict()
    try:
        import trimesh
    except ImportError:
        trimesh = None
    if trimesh is None:
        wanted = [m for m in ("watertight", "iou", "chamfer") if m in (getattr(clf.temp, "metrics", None) or [])]
        if wanted:
            print("WARNING: trimesh is not installed; mesh metrics {} are not computed for {}".format(wanted, getattr(data, "filename", "")))
        return InterfaceMesh(mdata["vertices"], faces), eval_dict
    recon_mesh = trimesh.Trimesh(mdata["vertices"], faces, process=True)
    if getattr(clf.temp, "fix_orientation", None):
        trimesh.repair.fix_normals(recon_mesh)
    metrics = getattr(clf.temp, "metrics", None) or []
    if "watertight" in metrics:
        eval_dict["watertight"] = int(recon_mesh.is_watertight)
    if "chamfer" in metrics:
        from scipy.spatial import cKDTree
        subfolder = data['id'] if data['id'] else data['category']
        gt_points = np.load(os.path.join(data.path, "eval", subfolder, "pointcloud.npz"))["points"].astype(np.float32)
        recon_points = recon_mesh.sample(gt_points.shape[0], return_index=False)
        d1, _ = cKDTree(recon_points).query(gt_points)
        d2, _ = cKDTree(gt_points).query(recon_points)
        eval_dict["chamfer"] = 0.5 * (float(d1.mean()) + float(d2.mean()))
    return recon_mesh, eval_dict
