"""Logits -> labels -> interface facets on the GPU (reference processing/generate_mesh.py:61-107).

``extract_interface(prediction, infinite, nfacets)`` reproduces what ``generate`` does between the network output
and the trimesh call: labels of the finite cells (:75), the infinite cell appended as OUTSIDE (:93-99) and the list
of facets whose two cells carry different labels (:101-105) -- two Python loops over all facets in the reference,
three small kernels here.  The optional integer alpha-expansion graph cut (:15-58, third-party ``gco``) and the
``trimesh`` mesh object are CPU-side third-party steps and stay outside; ``labels`` can be replaced by the
graph-cut labels before ``interface_from_labels``.
"""
from __future__ import annotations

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
