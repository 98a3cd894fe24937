"""Tensor-level wrappers over the C ABI (PyTorch is used for device memory and streams only).

Every function takes CUDA(ROCm) float32 tensors whose last dimension is contiguous, launches on the
current torch stream and returns torch tensors.  No CPU path exists: a CPU tensor raises.
"""
from __future__ import annotations

import torch

from ._lib import DgnnError, check, lib, on_device_of, ptr, stream_ptr

DGNN_E_UNSUPPORTED = -2  # include/dgnn_hip.h


def _req(t: torch.Tensor, name: str, dtype=torch.float32, dim=None, any_stride=False):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise DgnnError("%s is on %s: dgnn_amd runs on the GPU only (no CPU fallback)" % (name, t.device))
    if (t.dtype not in dtype) if isinstance(dtype, tuple) else (t.dtype != dtype):
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if dim is not None and t.dim() != dim:
        raise ValueError("%s must be %d-D, got shape %s" % (name, dim, tuple(t.shape)))
    if t.dim() == 2 and t.numel() and t.size(1) > 1 and t.stride(1) != 1 and not any_stride:   # (a single column has no column stride to speak of)
        raise ValueError("%s must have a contiguous last dimension (stride %s)" % (name, t.stride()))
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.size(0) > 1 or t.stride(0) >= t.size(1) else t.size(1)


def rows2d(t: torch.Tensor, name: str) -> torch.Tensor:
    """Returns `t` as a 2-D fp32 tensor with unit column stride (copies only when it must)."""
    _req(t, name, dim=2)
    return t


ACT = (torch.float32, torch.bfloat16)   # storage types of activations: fp32 (reference arithmetic) or bf16 (BASELINE config 3)


def _sfx(t: torch.Tensor) -> str:
    """entry-point suffix for the storage type of `t`"""
    return "_bf16" if t.dtype == torch.bfloat16 else ""


def _same(t, ref, name):
    if t is not None and t.dtype != ref.dtype:
        raise TypeError("%s must have the storage type of its companion (%s), got %s" % (name, ref.dtype, t.dtype))
    return t


def _f32(n, device):
    return torch.empty(int(max(n, 1)), dtype=torch.float32, device=device)


def _f32_work(n, device):
    """work buffer whose size changes from step to step (training blocks differ in size): rounded up to a multiple of 2M floats so
    that the caching allocator sees a handful of sizes instead of a new one every step (each miss is a hipMalloc, and those
    synchronise: the first ~300 steps of a training run were 40 % slower until its cache had filled)"""
    n = int(max(n, 1))
    return torch.empty((n + (1 << 21) - 1) >> 21 << 21, dtype=torch.float32, device=device)


# ---- plan ---------------------------------------------------------------------------------------
PLAN_HINT_AUTO, PLAN_HINT_GROUPED, PLAN_HINT_REFERENCE, PLAN_HINT_GENERIC = 0, 1, 2, 3
PLAN_HINT_GROUPED_TRUSTED = 4     # GROUPED without the device-side check / standby builder: for lists this package laid out itself (partition.build_ring_part)


@on_device_of
def plan_build(edge_index: torch.Tensor, n_key: int, by: int, hint: int = PLAN_HINT_AUTO, n_other: int = 0):
    """-> (rowptr int32 [n_key+1], other int32 [E], eid int32 [E]); see dgnn_plan_build.  `n_other`: node count of the other
    side for range checking (0 = unchecked)."""
    _req(edge_index, "edge_index", torch.int64, 2, any_stride=True)
    if edge_index.size(0) != 2:
        raise ValueError("edge_index must be an int64 [2,E] tensor (any strides; the transposed [E,2] view is read in place)")
    E = edge_index.size(1)
    dev = edge_index.device
    rowptr = torch.empty(n_key + 1, dtype=torch.int32, device=dev)
    other = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
    eid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
    scratch = torch.empty(int(lib().dgnn_plan_scratch_elems(E, n_key)), dtype=torch.int32, device=dev)
    check(lib().dgnn_plan_build(ptr(edge_index), edge_index.stride(0), edge_index.stride(1), E, n_key, int(n_other), by, hint, ptr(rowptr), ptr(other), ptr(eid), ptr(scratch), stream_ptr()),
          "dgnn_plan_build", poll=True)
    return rowptr, other, eid


@on_device_of
def gather_rows(src: torch.Tensor, idx: torch.Tensor, cols: int = None) -> torch.Tensor:
    """out = src[idx, :cols] (cols None: all columns); only the requested columns are read"""
    if src.dtype in (torch.bfloat16, torch.int16):   # rows of 16-bit pairs (bf16, or unsigned rows in their int16 container) move as fp32 words (bit copies)
        if not src.is_contiguous() or src.size(1) % 2 or (cols is not None and cols % 2):
            raise ValueError("16-bit gather_rows needs contiguous rows of even width")
        return gather_rows(src.view(torch.float32), idx, None if cols is None else cols // 2).view(src.dtype)
    _req(src, "src", dim=2)
    _req(idx, "idx", torch.int32, 1)
    cols = src.size(1) if cols is None else int(cols)
    if not 0 < cols <= src.size(1):
        raise ValueError("cols=%d outside (0, %d]" % (cols, src.size(1)))
    out = torch.empty((idx.numel(), cols), dtype=torch.float32, device=src.device)
    check(lib().dgnn_gather_rows_f32(ptr(src), _ld(src), ptr(idx), idx.numel(), cols, ptr(out), cols, stream_ptr()),
          "dgnn_gather_rows_f32")
    return out


@on_device_of
def scatter_rows_(out: torch.Tensor, idx: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    if src.dtype in (torch.bfloat16, torch.int16):   # bit copies of 16-bit pairs as fp32 words
        if not (src.is_contiguous() and out.is_contiguous()) or src.size(1) % 2 or out.dtype != src.dtype:
            raise ValueError("16-bit scatter_rows_ needs contiguous rows of even width and one row format")
        scatter_rows_(out.view(torch.float32), idx, src.view(torch.float32))
        return out
    _req(src, "src", dim=2)
    _req(out, "out", dim=2)
    _req(idx, "idx", torch.int64, 1)
    check(lib().dgnn_scatter_rows_f32(ptr(src), _ld(src), ptr(idx), idx.numel(), src.size(1), ptr(out), _ld(out), stream_ptr()),
          "dgnn_scatter_rows_f32")
    return out


@on_device_of
def relu(x: torch.Tensor) -> torch.Tensor:
    _req(x, "x", ACT)
    x = x.contiguous()
    y = torch.empty_like(x)
    check(getattr(lib(), "dgnn_relu" + _sfx(x))(ptr(x), x.numel(), ptr(y), stream_ptr()), "dgnn_relu")
    return y


@on_device_of
def relu_bwd(y: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    _req(y, "y", ACT)
    _same(_req(g, "g", ACT), y, "g")
    y, g = y.contiguous(), g.contiguous()
    out = torch.empty_like(g)
    check(getattr(lib(), "dgnn_relu_bwd" + _sfx(y))(ptr(y), ptr(g), g.numel(), ptr(out), stream_ptr()), "dgnn_relu_bwd")
    return out


# ---- aggregation --------------------------------------------------------------------------------
@on_device_of
def aggregate_fwd(rowptr, src, eid, n_dst, x_src, edge_attr=None, We=None, be=None, phi=None, want_phi=False):
    _req(x_src, "x_src", ACT, dim=2)
    c_in = x_src.size(1)
    a = torch.empty((n_dst, c_in), dtype=x_src.dtype, device=x_src.device)
    phi_out = None
    f_e = 0
    if We is not None:
        _req(edge_attr, "edge_attr", dim=2)
        _req(We, "We", dim=2)
        We = We.contiguous()
        be = _req(be, "be").contiguous()
        f_e = We.size(1)
        if We.size(0) != c_in or edge_attr.size(1) < f_e:
            raise ValueError("lin_e weight %s does not match c_in=%d / edge_attr %s" % (tuple(We.shape), c_in, tuple(edge_attr.shape)))
        if want_phi:
            phi_out = torch.empty((edge_attr.size(0), c_in), dtype=x_src.dtype, device=x_src.device)
    elif phi is not None:
        _same(_req(phi, "phi", ACT, dim=2), x_src, "phi")
    check(getattr(lib(), "dgnn_sage_aggregate_fwd" + _sfx(x_src))(
        ptr(rowptr), ptr(src), ptr(eid), n_dst, ptr(x_src), _ld(x_src), c_in,
        ptr(edge_attr) if We is not None else None, _ld(edge_attr) if We is not None else 0, f_e,
        ptr(We), ptr(be), ptr(phi), _ld(phi) if phi is not None else 0,
        ptr(phi_out), c_in, ptr(a), c_in, stream_ptr()), "dgnn_sage_aggregate_fwd")
    return (a, phi_out) if want_phi else a


@on_device_of
def aggregate_bwd(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, da, edge_attr=None, We=None, be=None, phi=None,
                  need_dx=True):
    """-> (dx_src | None, dWe | None, dbe | None, dphi | None)"""
    _req(x_src, "x_src", ACT, dim=2)
    _same(_req(da, "da", ACT, dim=2), x_src, "da")
    _same(phi, x_src, "phi")
    c_in = x_src.size(1)
    dev = x_src.device
    dx = torch.empty((n_src, c_in), dtype=x_src.dtype, device=dev) if need_dx else None
    dWe = dbe = dphi = None
    f_e = 0
    partials = None
    if We is not None:
        We = We.contiguous()
        be = be.contiguous()
        f_e = We.size(1)
        dWe = torch.empty_like(We)     # written, not accumulated, by the slab reduction
        dbe = torch.empty_like(be)
        partials = _f32(lib().dgnn_sage_aggregate_bwd_scratch_elems(n_src, c_in, f_e), dev)
    elif phi is not None:
        dphi = torch.zeros((phi.size(0), c_in), dtype=x_src.dtype, device=dev)
    check(getattr(lib(), "dgnn_sage_aggregate_bwd" + _sfx(x_src))(
        ptr(t_rowptr), ptr(t_dst), ptr(t_eid), n_src, ptr(rowptr_dst), ptr(x_src), _ld(x_src), c_in,
        ptr(edge_attr) if We is not None else None, _ld(edge_attr) if We is not None else 0, f_e, ptr(We), ptr(be),
        ptr(phi), _ld(phi) if phi is not None else 0, ptr(da), _ld(da), ptr(dx), c_in, ptr(dWe), ptr(dbe), ptr(dphi), c_in,
        ptr(partials), stream_ptr()), "dgnn_sage_aggregate_bwd")
    return dx, dWe, dbe, dphi


# ---- dense --------------------------------------------------------------------------------------
@on_device_of
def linear_fwd(A1, W1, A2=None, W2=None, bias=None, scale=None, shift=None, relu=False, out=None, out_dtype=None):
    """out = act((A1.W1^T + A2.W2^T + bias) * scale + shift).  fp32 operands: fp32 MFMA, fp32 out.  bf16 operands (bf16 storage
    path): bf16 MFMA with fp32 accumulation, `out_dtype` bf16 (default) or fp32 (logits); an fp32 A next to bf16 storage (a
    gradient of fp32 logits) is rounded to bf16 first."""
    _req(A1, "A1", ACT, dim=2)
    W1 = _req(W1, "W1", dim=2).contiguous()
    M, n_out = A1.size(0), W1.size(0)
    if W1.size(1) != A1.size(1):
        raise ValueError("W1 %s does not match A1 %s" % (tuple(W1.shape), tuple(A1.shape)))
    if A2 is not None:
        _same(_req(A2, "A2", ACT, dim=2), A1, "A2")
        W2 = _req(W2, "W2", dim=2).contiguous()
        if A2.size(0) != M or W2.size(0) != n_out or W2.size(1) != A2.size(1):
            raise ValueError("second operand shapes do not match")
    bf = A1.dtype == torch.bfloat16
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else A1.dtype
    if not bf and out_dtype != torch.float32:
        raise TypeError("fp32 operands produce fp32 output")
    if out is None:
        out = torch.empty((M, n_out), dtype=out_dtype, device=A1.device)
    elif out.size(0) < M or out.size(1) != n_out or out.stride(0) != n_out or out.dtype != out_dtype:
        raise ValueError("out must be a contiguous [>= M, n_out] buffer of the output type")
    if bf:
        check(lib().dgnn_linear_fwd_bf16(
            ptr(A1), _ld(A1), A1.size(1), ptr(W1), W1.size(1),
            ptr(A2), _ld(A2) if A2 is not None else 0, A2.size(1) if A2 is not None else 0, ptr(W2), W2.size(1) if W2 is not None else 0,
            ptr(bias), ptr(scale), ptr(shift), int(bool(relu)), M, n_out, ptr(out), n_out, int(out_dtype == torch.float32), stream_ptr()),
            "dgnn_linear_fwd_bf16")
        return out
    # fp32 operands: bit-faithful fp32 MFMA ("f32" mode) or the fused layers' exact-split bf16 arithmetic (fp32-class, 6/16 of the time)
    if GEMM_MODE == GEMM_F16X2 and M >= 8192 and n_out > 256:  # (at n_out <= 256 the extra pass over A for the row scales costs what the products save)
        # the wide conv layers in the fused layers' fp16 two-part form (X2HP: see above)
        k1, k2 = A1.size(1), (A2.size(1) if A2 is not None else 0)
        if X2HP:
            scratch = torch.empty(int(lib().dgnn_linear_fwd_x2hp_scratch_elems(M, n_out, k1, k2)), dtype=torch.float32, device=A1.device)
            entry, name = lib().dgnn_linear_fwd_x2hp, "dgnn_linear_fwd_x2hp"
        else:
            scratch = torch.empty(int(lib().dgnn_linear_fwd_x2h_scratch_elems(M, n_out)), dtype=torch.float32, device=A1.device)
            entry, name = lib().dgnn_linear_fwd_x2h, "dgnn_linear_fwd_x2h"
        rc = entry(
            ptr(A1), _ld(A1), k1, ptr(W1), W1.size(1),
            ptr(A2), _ld(A2) if A2 is not None else 0, k2, ptr(W2), W2.size(1) if W2 is not None else 0,
            ptr(bias), ptr(scale), ptr(shift), int(bool(relu)), M, n_out, ptr(out), n_out, ptr(scratch), stream_ptr())
        if rc != DGNN_E_UNSUPPORTED:
            check(rc, name)
            return out
    fn = lib().dgnn_linear_fwd if GEMM_MODE == GEMM_F32 else lib().dgnn_linear_fwd_x3
    check(fn(
        ptr(A1), _ld(A1), A1.size(1), ptr(W1), W1.size(1),
        ptr(A2), _ld(A2) if A2 is not None else 0, A2.size(1) if A2 is not None else 0, ptr(W2), W2.size(1) if W2 is not None else 0,
        ptr(bias), ptr(scale), ptr(shift), int(bool(relu)), M, n_out, ptr(out), n_out, stream_ptr()), "dgnn_linear_fwd")
    return out


@on_device_of
def linear_wgrad(A, B):
    """dW[n_a,n_b] = A^T . B  (A [M,n_a], B [M,n_b])"""
    _req(A, "A", ACT, dim=2)
    _req(B, "B", ACT, dim=2)
    M, na, nb = A.size(0), A.size(1), B.size(1)
    dW = torch.empty((na, nb), dtype=torch.float32, device=A.device)
    if M == 0:
        return dW.zero_()
    partials = _f32(lib().dgnn_linear_wgrad_scratch_elems(M, na, nb), A.device)
    if A.dtype == torch.bfloat16 or B.dtype == torch.bfloat16:   # bf16 storage path: bf16 MFMA, fp32 gradient
        check(lib().dgnn_linear_wgrad_bf16(ptr(A), int(A.dtype == torch.float32), _ld(A), na, ptr(B), int(B.dtype == torch.float32), _ld(B), nb,
                                           M, ptr(dW), nb, 0, ptr(partials), stream_ptr()), "dgnn_linear_wgrad_bf16")
        return dW
    fn = lib().dgnn_linear_wgrad if GEMM_MODE == GEMM_F32 else lib().dgnn_linear_wgrad_x3
    check(fn(ptr(A), _ld(A), na, ptr(B), _ld(B), nb, M, ptr(dW), nb, 0, ptr(partials), stream_ptr()), "dgnn_linear_wgrad")
    return dW


@on_device_of
def linear_fwd_with_batch_stats(A1, W1, A2=None, W2=None, bias=None, gamma=None, beta=None, eps=1e-5):
    """z = A1 . W1^T + A2 . W2^T + bias and the BatchNorm batch statistics of z from the GEMM's own epilogue (dgnn_linear_fwd_x3_stats +
    dgnn_bn_stats_finalize_fold) -> (z, mean, var, scale, shift), or None where the GEMM takes the tile without that epilogue."""
    _req(A1, "A1", dim=2)
    M, n_out = A1.size(0), W1.size(0)
    z = torch.empty((M, n_out), dtype=torch.float32, device=A1.device)
    scratch = _f32(lib().dgnn_colstats_scratch_elems(M, n_out) + 2, A1.device)
    cs = (scratch.data_ptr() + 7) & ~7
    W1 = W1.contiguous()
    W2 = W2.contiguous() if W2 is not None else None
    rc = lib().dgnn_linear_fwd_x3_stats(ptr(A1), _ld(A1), A1.size(1), ptr(W1), W1.size(1), ptr(A2), _ld(A2) if A2 is not None else 0,
                                        A2.size(1) if A2 is not None else 0, ptr(W2), W2.size(1) if W2 is not None else 0, ptr(bias), M, n_out, ptr(z), n_out,
                                        cs, stream_ptr())
    if rc == DGNN_E_UNSUPPORTED:
        return None
    check(rc, "dgnn_linear_fwd_x3_stats")
    st = torch.empty((4, n_out), dtype=torch.float32, device=A1.device)
    check(lib().dgnn_bn_stats_finalize_fold(cs, (M + 31) // 32, M, n_out, ptr(st[0]), ptr(st[1]), None, None, 0.0, ptr(gamma), ptr(beta), eps, ptr(st[2]),
                                            ptr(st[3]), stream_ptr()), "dgnn_bn_stats_finalize_fold")
    return z, st[0], st[1], st[2], st[3]


@on_device_of
def linear_wgrad_cat(A, B1, B2=None, bias=True):
    """(dW1 = A^T . B1, dW2 = A^T . B2 | None, column sums of A | None) from one launch pair (dgnn_linear_wgrad_x3_cat; fp32, x3 arithmetic)"""
    _req(A, "A", dim=2)
    _req(B1, "B1", dim=2)
    M, na, nb1 = A.size(0), A.size(1), B1.size(1)
    nb2 = B2.size(1) if B2 is not None else 0
    dW1 = torch.empty((na, nb1), dtype=torch.float32, device=A.device)
    dW2 = torch.empty((na, nb2), dtype=torch.float32, device=A.device) if B2 is not None else None
    db = torch.empty(na, dtype=torch.float32, device=A.device) if bias else None
    scratch = _f32(lib().dgnn_linear_wgrad_cat_scratch_elems(M, na, nb1, nb2), A.device)
    check(lib().dgnn_linear_wgrad_x3_cat(ptr(A), _ld(A), na, ptr(B1), _ld(B1), nb1, ptr(B2), _ld(B2) if B2 is not None else 0, nb2, M, ptr(dW1), ptr(dW2),
                                         ptr(db), ptr(scratch), stream_ptr()), "dgnn_linear_wgrad_x3_cat")
    return dW1, dW2, db


@on_device_of
def aggregate_bwd_add(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x_src, da, edge_attr, We, be, add):
    """aggregate_bwd (fused 20-attribute filter) with dx[:add.size(0)] += add folded into the store -> (dx, dWe, dbe)"""
    _req(x_src, "x_src", dim=2)
    c_in = x_src.size(1)
    dx = torch.empty((n_src, c_in), dtype=torch.float32, device=x_src.device)
    We, be = We.contiguous(), be.contiguous()
    dWe, dbe = torch.empty_like(We), torch.empty_like(be)
    partials = _f32(lib().dgnn_sage_aggregate_bwd_scratch_elems(n_src, c_in, We.size(1)), x_src.device)
    check(lib().dgnn_sage_aggregate_bwd_add(ptr(t_rowptr), ptr(t_dst), ptr(t_eid), n_src, ptr(rowptr_dst), ptr(x_src), _ld(x_src), c_in, ptr(edge_attr),
                                            _ld(edge_attr), We.size(1), ptr(We), ptr(be), ptr(da), _ld(da), ptr(dx), c_in, ptr(add), _ld(add), add.size(0),
                                            ptr(dWe), ptr(dbe), ptr(partials), stream_ptr()), "dgnn_sage_aggregate_bwd_add")
    return dx, dWe, dbe


@on_device_of
def colsum(x):
    _req(x, "x", ACT, dim=2)
    out = torch.empty(x.size(1), dtype=torch.float32, device=x.device)
    if x.size(0) == 0:
        return out.zero_()
    scratch = _f32(lib().dgnn_colstats_scratch_elems(x.size(0), x.size(1)), x.device)
    check(getattr(lib(), "dgnn_colsum" + _sfx(x))(ptr(x), _ld(x), x.size(0), x.size(1), ptr(out), 0, ptr(scratch), stream_ptr()), "dgnn_colsum")
    return out


# ---- batch norm ---------------------------------------------------------------------------------
@on_device_of
def bn_fold(gamma, beta, mean, var, eps):
    c = mean.numel()
    scale = torch.empty(c, dtype=torch.float32, device=mean.device)
    shift = torch.empty(c, dtype=torch.float32, device=mean.device)
    check(lib().dgnn_bn_fold(ptr(gamma), ptr(beta), ptr(mean), ptr(var), float(eps), c, ptr(scale), ptr(shift), stream_ptr()),
          "dgnn_bn_fold")
    return scale, shift


def _written_behind_torch(*tensors):
    """the library wrote these through raw addresses: bump their version counters so that anything cached against them (the eval-mode BatchNorm
    folds, prepared parameters) is rebuilt"""
    ts = [t for t in tensors if t is not None]
    if ts:
        torch.autograd.graph.increment_version(ts)


@on_device_of
def bn_batch_stats(x, running_mean=None, running_var=None, momentum=0.1):
    _req(x, "x", ACT, dim=2)
    M, c = x.shape
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    var = torch.empty(c, dtype=torch.float32, device=x.device)
    scratch = _f32(lib().dgnn_colstats_scratch_elems(M, c), x.device)
    check(getattr(lib(), "dgnn_bn_batch_stats" + _sfx(x))(ptr(x), _ld(x), M, c, ptr(mean), ptr(var), ptr(running_mean), ptr(running_var),
                                    float(momentum), ptr(scratch), stream_ptr()), "dgnn_bn_batch_stats")
    _written_behind_torch(running_mean, running_var)
    return mean, var


@on_device_of
def scale_shift_act(x, scale, shift, relu):
    _req(x, "x", ACT, dim=2)
    y = torch.empty((x.size(0), x.size(1)), dtype=x.dtype, device=x.device)
    check(getattr(lib(), "dgnn_scale_shift_act" + _sfx(x))(ptr(x), _ld(x), ptr(scale), ptr(shift), int(bool(relu)), x.size(0), x.size(1), ptr(y),
                                     x.size(1), stream_ptr()), "dgnn_scale_shift_act")
    return y


@on_device_of
def bn_relu_bwd(x, y, dy, gamma, mean, var, eps, train, relu):
    _req(x, "x", ACT, dim=2)
    _same(_req(dy, "dy", ACT, dim=2), x, "dy")
    _same(y, x, "y")
    M, c = x.shape
    dx = torch.empty((M, c), dtype=x.dtype, device=x.device)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    scratch = _f32(lib().dgnn_colstats_scratch_elems(M, c), x.device)
    check(getattr(lib(), "dgnn_bn_relu_bwd" + _sfx(x))(ptr(x), _ld(x), ptr(y), _ld(y) if y is not None else 0, ptr(dy), _ld(dy), ptr(gamma), ptr(mean),
                                 ptr(var), float(eps), int(bool(train)), int(bool(relu)), M, c, ptr(dx), c, ptr(dgamma), ptr(dbeta),
                                 ptr(scratch), stream_ptr()), "dgnn_bn_relu_bwd")
    return dx, dgamma, dbeta


@on_device_of
def bn_relu_bwd_sums(x, y, dy, mean, var, eps, relu):
    """local [2, c] = (sum g, sum g * x_hat), g = dy behind the ReLU mask of y -- the half of bn_relu_bwd a partitioned scene all-reduces"""
    _req(x, "x", dim=2)
    M, c = x.shape
    sums = torch.empty((2, c), dtype=torch.float32, device=x.device)
    scratch = _f32(lib().dgnn_colstats_scratch_elems(M, c), x.device)
    check(lib().dgnn_bn_relu_bwd_sums(ptr(x), _ld(x), ptr(y), _ld(y) if y is not None else 0, ptr(dy), _ld(dy), ptr(mean), ptr(var), float(eps),
                                      int(bool(relu)), M, c, ptr(sums), ptr(scratch), stream_ptr()), "dgnn_bn_relu_bwd_sums")
    return sums


@on_device_of
def bn_relu_bwd_apply(x, y, dy, gamma, mean, var, eps, relu, sums, count):
    """dx of y = relu(bn(x)) from sums [2, c] taken over `count` rows (all ranks' rows of a partitioned scene)"""
    _req(x, "x", dim=2)
    M, c = x.shape
    dx = torch.empty((M, c), dtype=torch.float32, device=x.device)
    check(lib().dgnn_bn_relu_bwd_apply(ptr(x), _ld(x), ptr(y), _ld(y) if y is not None else 0, ptr(dy), _ld(dy), ptr(gamma), ptr(mean), ptr(var), float(eps),
                                       int(bool(relu)), M, c, ptr(sums.contiguous()), float(count), ptr(dx), c, stream_ptr()), "dgnn_bn_relu_bwd_apply")
    return dx


# ---- wide conv layers on split rows (csrc/wide.hip) ---------------------------------------------
# DGNN_WIDE_SR=0: the layers wider than the fused kernels keep the fp32 aggregate + x2h / x3 GEMM pair of rounds 2-4
WIDE_SR = __import__("os").environ.get("DGNN_WIDE_SR", "1") != "0"


class SplitRows:
    """A wide activation [n, C] as the wide kernels store it (include/dgnn_hip.h "SPLIT ROWS"): `data` uint8 [n, C * 4] -- per 32 channels a 128-byte chunk
    [hi x 32 | lo x 32] fp16 of x * s in the chunk's position order -- and `scales` fp32 [n, ceil(C / 256)], one power of two per row and 256 channels.
    Produced by aggregate_sr / linear_sr / pack_rows; `.float()` gives the fp32 rows back (22 significand bits per value)."""

    def __init__(self, data, scales, channels):
        self.data, self.scales, self.channels = data, scales, int(channels)

    size = lambda self, d=None: (self.data.size(0), self.channels) if d is None else (self.data.size(0), self.channels)[d]
    shape = property(lambda self: (self.data.size(0), self.channels))
    device = property(lambda self: self.data.device)
    dtype = "split_rows"
    is_cuda = property(lambda self: self.data.is_cuda)

    def __getitem__(self, sl):
        """row ranges only (a prefix / sub-range of the rows): views of both tensors"""
        if not isinstance(sl, slice) or sl.step not in (None, 1):
            raise TypeError("SplitRows supports contiguous row ranges only")
        return SplitRows(self.data[sl], self.scales[sl], self.channels)

    def float(self):
        return unpack_rows(self)


def sr_groups(c):
    return (int(c) + 255) // 256


@on_device_of
def pack_rows(A1, A2=None, per_row=False):
    """fp32 rows [n, k1] (| [n, k2]) -> SplitRows of [A1 | A2] (widths multiples of 32); per_row: ONE scale per row (weights) instead of one per 256 channels"""
    _req(A1, "A1", dim=2)
    n, k1 = A1.shape
    k2 = 0
    if A2 is not None:
        _req(A2, "A2", dim=2)
        k2 = A2.size(1)
    C_ = k1 + k2
    if k1 % 32 or k2 % 32:
        raise ValueError("split rows need widths that are multiples of 32")
    ng = 1 if per_row else max(1, (C_ // 32 + 7) // 8)
    data = torch.empty((n, C_ * 4), dtype=torch.uint8, device=A1.device)
    scales = torch.empty((n, ng), dtype=torch.float32, device=A1.device)
    check(lib().dgnn_sr_pack(ptr(A1), _ld(A1), k1, ptr(A2), _ld(A2) if A2 is not None else 0, k2, n, 0 if per_row else 8, ptr(data), C_ * 4, ptr(scales), ng,
                             stream_ptr()), "dgnn_sr_pack")
    return SplitRows(data, scales, C_)


@on_device_of
def unpack_rows(sr: "SplitRows") -> torch.Tensor:
    n, C_ = sr.size()
    out = torch.empty((n, C_), dtype=torch.float32, device=sr.device)
    ng = sr.scales.size(1)
    gch = 8 if ng == sr_groups(C_) and not (ng == 1 and C_ > 256) else 0
    check(lib().dgnn_sr_unpack(ptr(sr.data), sr.data.stride(0), ptr(sr.scales), ng, gch, C_, n, ptr(out), C_, stream_ptr()), "dgnn_sr_unpack")
    return out


def wide_layer_supported(c_in: int, c_out: int, f_e: int) -> bool:
    """dgnn_sage_aggregate_sr + dgnn_linear_sr take this conv layer (default arithmetic only)"""
    return WIDE_SR and GEMM_MODE == GEMM_F16X2 and f_e == 20 and c_in in (128, 256, 512) and c_out % 256 == 0 and c_out > 0


@on_device_of
def sr_prepare_filter(We, be):
    c = We.size(0)
    nb = int(lib().dgnn_sr_filter_prepared_bytes(c))
    if nb <= 0 or We.size(1) != 20:
        return None
    buf = torch.empty(nb, dtype=torch.uint8, device=We.device)
    check(lib().dgnn_sr_prepare_filter(ptr(We.contiguous()), ptr(be.contiguous()), c, ptr(buf), stream_ptr()), "dgnn_sr_prepare_filter")
    return buf


@on_device_of
def aggregate_sr(rowptr, src, eid, n_dst, x, edge_attr, We, be, prep, own_rows=False):
    """-> a (SplitRows [n_dst, C]) [, x[:n_dst] as SplitRows when `own_rows` and x is an fp32 tensor] or None when the library declines the layout"""
    sr_in = isinstance(x, SplitRows)
    c = x.size(1)
    dev = x.device
    ng = sr_groups(c)
    a = SplitRows(torch.empty((n_dst, c * 4), dtype=torch.uint8, device=dev), torch.empty((n_dst, ng), dtype=torch.float32, device=dev), c)
    xo = None
    if own_rows and not sr_in:
        xo = SplitRows(torch.empty((n_dst, c * 4), dtype=torch.uint8, device=dev), torch.empty((n_dst, ng), dtype=torch.float32, device=dev), c)
    if sr_in:
        xp, ldx, xs = ptr(x.data), x.data.stride(0), ptr(x.scales)
    else:
        _req(x, "x", dim=2)
        xp, ldx, xs = ptr(x), _ld(x), None
    rc = lib().dgnn_sage_aggregate_sr(ptr(rowptr), ptr(src), ptr(eid), n_dst, xp, int(sr_in), ldx, xs, c, ptr(edge_attr), _ld(edge_attr), ptr(We.contiguous()),
                                      ptr(be.contiguous()), ptr(prep), ptr(a.data), ptr(a.scales), ptr(xo.data) if xo is not None else None,
                                      ptr(xo.scales) if xo is not None else None, stream_ptr())
    if rc == DGNN_E_UNSUPPORTED:
        return None
    check(rc, "dgnn_sage_aggregate_sr")
    return (a, xo) if own_rows else a


@on_device_of
def linear_sr(A1, Wp, A2=None, bias=None, scale=None, shift=None, relu=False, out_f32=False, rows=None, proj=None):
    """act(([A1 | A2] . W^T + bias) * scale + shift): A1 / A2 SplitRows, Wp = pack_rows(W1, W2, per_row=True) -> SplitRows; with out_f32 an fp32 tensor;
    with proj = (W3 [n_proj, n_out], b3 | None) the logits act(...) . W3^T + b3 [M, n_proj] (the decoder's output Linear inside the launch).
    `rows`: only the first `rows` rows of the operands.  None when the library declines the shape."""
    M = A1.size(0) if rows is None else int(rows)
    n_out = Wp.size(0)
    dev = A1.device
    c1, c2 = A1.channels, (A2.channels if A2 is not None else 0)
    if Wp.channels != c1 + c2:
        raise ValueError("packed weights have %d input channels, operands %d" % (Wp.channels, c1 + c2))
    o32 = osr = lg = W3 = b3 = None
    n_proj = 0
    if proj is not None:
        W3, b3 = proj
        W3 = _req(W3, "W3", dim=2).contiguous()
        n_proj = W3.size(0)
        if W3.size(1) != n_out or n_proj not in (1, 2):
            return None
        lg = (torch.zeros if n_out > 256 else torch.empty)((M, n_proj), dtype=torch.float32, device=dev)     # (column tiles add into zeroed logits)
    elif out_f32:
        o32 = torch.empty((M, n_out), dtype=torch.float32, device=dev)
    else:
        osr = SplitRows(torch.empty((M, n_out * 4), dtype=torch.uint8, device=dev), torch.empty((M, sr_groups(n_out)), dtype=torch.float32, device=dev), n_out)
    rc = lib().dgnn_linear_sr(ptr(A1.data), A1.data.stride(0), ptr(A1.scales), c1, ptr(A2.data) if A2 is not None else None,
                              A2.data.stride(0) if A2 is not None else 0, ptr(A2.scales) if A2 is not None else None, c2, ptr(Wp.data), ptr(Wp.scales),
                              ptr(bias), ptr(scale), ptr(shift), int(bool(relu)), M, n_out, ptr(osr.data) if osr is not None else None,
                              n_out * 4 if osr is not None else 0, ptr(osr.scales) if osr is not None else None, ptr(o32), n_out, ptr(W3), ptr(b3), n_proj, ptr(lg),
                              stream_ptr())
    if rc == DGNN_E_UNSUPPORTED:
        return None
    check(rc, "dgnn_linear_sr")
    return lg if proj is not None else (o32 if out_f32 else osr)


# ---- fused inference layer ----------------------------------------------------------------------
def fused_layer_supported(c_in: int, c_out: int, f_e: int, x: torch.Tensor = None) -> bool:
    """Mirrors the shape dispatch of dgnn_sage_layer_fused_fwd (csrc/fused.hip): c_in <= 64 -> c_out in {64,128};
    64 < c_in <= 128 -> c_out == 128 with even c_in / row stride and 8-byte aligned rows.  Everything else takes the
    aggregate + linear pair."""
    if not FUSED_ENABLED or f_e != 20:
        return False
    if c_in <= 64:
        return c_out in (64, 128)
    if c_in <= 128:
        ok = c_out == 128 and c_in % 2 == 0
        if x is not None:
            ok = ok and x.stride(0) % 2 == 0 and x.data_ptr() % 8 == 0
        return ok
    return False


FUSED_ENABLED = True
GEMM_F32, GEMM_BF16X3, GEMM_BF16X3_FILTER, GEMM_F16X2_DENSE, GEMM_F16X2 = 0, 1, 2, 3, 4
# wide GEMMs with operands pre-split by a pass of their own and staged by DMA (dgnn_linear_fwd_x2hp, bit-identical to x2h).  OFF by default: the GEMM
# itself runs 1.3-1.5x faster (317 vs 200-260 TFLOP/s fp32-equivalent at M = 1M) but the extra write + read of the split operand costs more
# than that saves (512 -> 1024 layer 11.7 vs 10.8 ms, 256 -> 512 4.8 vs 4.2 ms; profiles/r03_wide.md)
X2HP = __import__("os").environ.get("DGNN_X2HP", "0") != "0"
GEMM_MODE_NAMES = {"f32": GEMM_F32, "bf16x3": GEMM_BF16X3, "bf16x3f": GEMM_BF16X3_FILTER, "f16x2d": GEMM_F16X2_DENSE, "f16x2": GEMM_F16X2}
# how the fused layer uses the matrix cores: exact-fp32 MFMA for the dense part ("f32"), 3-way split-bf16 MFMA for the
# dense part ("bf16x3"), or split-bf16 MFMA for the dense part AND the filter MLP ("bf16x3f"); all fp32-class accuracy
# "f16x2d" / "f16x2": the dense product (and the filter MLP) on fp16 x 2 with power-of-two group scales (22 significand bits, 3 products:
# the 3xTF32 scheme) -- half the matrix work of the bf16 x 3 forms; GEMM entry points outside the fused layer keep bf16 x 3
GEMM_MODE = GEMM_MODE_NAMES[__import__("os").environ.get("DGNN_GEMM_MODE", "f16x2")]
# whole-graph inference: fused layers read the caller's edge_attr in place (rows DMA-gathered by the plan's eid) instead of
# staging a plan-ordered copy once per scene; DGNN_EDGE_STAGING=1 restores the staged copy
EDGE_GATHER_IN_KERNEL = __import__("os").environ.get("DGNN_EDGE_STAGING", "0") != "1"


FUSED_MAX_ELEMS = 1 << 31  # row offsets inside one fused-layer launch are 32-bit element counts (tests lower it)

# optional profiling hook (bench.py): SurfaceNet._eval_layers calls tok = hook(None, c_in, c_out, n_dst) right before the
# launches of one eval-mode conv layer (fused: one launch; other widths: aggregate + GEMM) and hook(tok, c_in, c_out, n_dst)
# right after them, on the launching thread / current stream
LAYER_HOOK = None


@on_device_of
def sage_layer_fused_fwd(rowptr, src, n_dst, x_src, edge_attr, We, be, Wj, bj, Wi, scale, shift, relu, gemm_mode=None,
                         out=None, eid=None, x_dst=None, prepared=None):
    """`x_dst` (optional): own rows of the destinations when they are not x_src[:n_dst] (a destination sub-range).
    `eid` (the plan's edge ids): edge_attr is the caller's tensor in its own row order and rows are gathered inside the
    kernel; eid=None: edge_attr is already in plan order.
    `out` (optional): a [>= n_dst, c_out] buffer whose first n_dst rows receive the result (the partitioned
    forward passes the next layer's [n_own + n_halo, C] activation buffer, so no copy is needed)."""
    _req(x_src, "x_src", dim=2)
    c_in, c_out = x_src.size(1), Wj.size(0)
    if out is None:
        out = torch.empty((n_dst, c_out), dtype=torch.float32, device=x_src.device)
    else:
        _req(out, "out", dim=2)
        if out.size(0) < n_dst or out.size(1) != c_out or out.stride(0) != c_out:
            raise ValueError("out must be a contiguous [>= n_dst, c_out] buffer")
    mode = GEMM_MODE if gemm_mode is None else gemm_mode
    if prepared is not None and mode == GEMM_F16X2:
        # `prepared` (sage_layer_prepare): the launch skips its parameter prologue; a shape the prepared kernels do not take falls through
        rc = lib().dgnn_sage_layer_fused_fwd_p(
            ptr(rowptr), ptr(src), ptr(eid), n_dst, ptr(x_src), ptr(x_dst), _ld(x_src), c_in, ptr(edge_attr), _ld(edge_attr), We.size(1),
            ptr(We), ptr(be), ptr(Wj), ptr(bj), ptr(Wi), ptr(scale), ptr(shift), int(bool(relu)), c_out, ptr(out), c_out, ptr(prepared), stream_ptr())
        if rc != DGNN_E_UNSUPPORTED:
            check(rc, "dgnn_sage_layer_fused_fwd_p")
            return out
    check(lib().dgnn_sage_layer_fused_fwd(
        ptr(rowptr), ptr(src), ptr(eid), n_dst, ptr(x_src), ptr(x_dst), _ld(x_src), c_in, ptr(edge_attr), _ld(edge_attr), We.size(1),
        ptr(We), ptr(be), ptr(Wj), ptr(bj), ptr(Wi), ptr(scale), ptr(shift), int(bool(relu)), c_out, ptr(out), c_out,
        mode, stream_ptr()), "dgnn_sage_layer_fused_fwd")
    return out


# parameters of the fused layers prepared once per set of weights (dgnn_sage_layer_prepare); DGNN_PREPARED=0: every launch derives them itself
PREPARED_PARAMS = __import__("os").environ.get("DGNN_PREPARED", "1") != "0"


@on_device_of
def sage_layer_prepare(We, be, Wj, Wi, decoder=None):
    """-> uint8 buffer for `prepared=` of sage_layer_fused_fwd / sage_layer_fused_decoder_fwd, or None when the shape has no prepared form.
    `decoder` = (W0, b0, scale1, shift1, W3, b3) for the last layer's launch that carries the decoder."""
    c_in, c_out = Wj.size(1), Wj.size(0)
    nbytes = int(lib().dgnn_sage_layer_prepared_bytes(c_in, c_out, int(decoder is not None)))
    if nbytes <= 0 or We.size(1) != 20:
        return None
    buf = torch.empty(nbytes, dtype=torch.uint8, device=Wj.device)
    d = decoder if decoder is not None else (None,) * 6
    rc = lib().dgnn_sage_layer_prepare(c_in, c_out, ptr(We.contiguous()), ptr(be), ptr(Wj.contiguous()), ptr(Wi.contiguous()),
                                       ptr(d[0].contiguous() if d[0] is not None else None), ptr(d[1]), ptr(d[2]), ptr(d[3]),
                                       ptr(d[4].contiguous() if d[4] is not None else None), ptr(d[5]), ptr(buf), stream_ptr())
    if rc == DGNN_E_UNSUPPORTED:
        return None
    check(rc, "dgnn_sage_layer_prepare")
    return buf


# the last conv layer and the decoder in one launch (dgnn_sage_layer_fused_decoder_fwd); DGNN_FUSE_DECODER=0 keeps them apart
FUSE_DECODER = __import__("os").environ.get("DGNN_FUSE_DECODER", "1") != "0"


def fused_layer_decoder_supported(c_in: int, c_out: int, f_e: int, hidden: int, n_out: int, x=None) -> bool:
    """shipped shape only: 64 < c_in <= 128, c_out 128, 20 edge attributes, decoder 128 -> 64 -> 2, default arithmetic, fp32 rows"""
    ok = (FUSE_DECODER and FUSED_ENABLED and GEMM_MODE == GEMM_F16X2 and 64 < c_in <= 128 and c_in % 8 == 0 and c_out == 128 and f_e == 20
          and hidden == 64 and n_out == 2)
    if ok and x is not None:
        ok = x.dtype == torch.float32 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
    return ok


@on_device_of
def sage_layer_fused_decoder_fwd(rowptr, src, n_dst, x_src, edge_attr, We, be, Wj, bj, Wi, scale, shift, relu, W0, b0, scale1, shift1, W3, b3,
                                 out=None, eid=None, x_dst=None, prepared=None):
    """Last conv layer + decoder, one launch -> logits [n_dst, 2] (written into `out` when given: a contiguous [>= n_dst, 2] fp32 buffer)."""
    _req(x_src, "x_src", dim=2)
    if out is None:
        out = torch.empty((n_dst, 2), dtype=torch.float32, device=x_src.device)
    else:
        _req(out, "out", dim=2)
        if out.size(0) < n_dst or out.size(1) != 2 or out.stride(0) != 2:
            raise ValueError("out must be a contiguous [>= n_dst, 2] buffer")
    head = (ptr(rowptr), ptr(src), ptr(eid), n_dst, ptr(x_src), ptr(x_dst), _ld(x_src), x_src.size(1), ptr(edge_attr), _ld(edge_attr), We.size(1),
            ptr(We), ptr(be), ptr(Wj), ptr(bj), ptr(Wi), ptr(scale), ptr(shift), int(bool(relu)), Wj.size(0), ptr(W0.contiguous()), ptr(b0),
            ptr(scale1), ptr(shift1), W0.size(0), ptr(W3.contiguous()), ptr(b3), W3.size(0), ptr(out))
    if prepared is not None:
        check(lib().dgnn_sage_layer_fused_decoder_fwd_p(*head, ptr(prepared), stream_ptr()), "dgnn_sage_layer_fused_decoder_fwd_p")
    else:
        check(lib().dgnn_sage_layer_fused_decoder_fwd(*head, stream_ptr()), "dgnn_sage_layer_fused_decoder_fwd")
    return out


# whole-scene inference as ONE library call (dgnn_static_infer_fwd: plan + every conv layer + decoder); DGNN_INFER_ONE_CALL=0 keeps the per-layer calls
INFER_ONE_CALL = __import__("os").environ.get("DGNN_INFER_ONE_CALL", "1") != "0"


def _infer_tables(x_width, layers, decoder, prepared, cache):
    """ctypes argument tables of the one-call entry points: (L, widths, 7 per-layer pointer arrays, prepared, the decoder's 8 scalars / addresses,
    n_out); kept in `cache` (the model's, valid as long as the tensors are: SurfaceNet._one_call_tables) when one is given."""
    import ctypes as C
    if cache is not None:
        hit = cache.get(x_width)
        if hit is not None:
            return hit
    L = len(layers)
    widths = [x_width] + [l[2].size(0) for l in layers]
    w_arr = (C.c_int32 * (L + 1))(*widths)
    cols = tuple((C.c_void_p * L)(*[(l[i].data_ptr() if l[i] is not None else None) for l in layers]) for i in range(7))
    prep = (C.c_void_p * L)(*[(p.data_ptr() if p is not None else None) for p in prepared]) if prepared is not None else None
    d = decoder if decoder is not None else (None,) * 6
    dec = (ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), d[0].size(0) if d[0] is not None else 0, ptr(d[4]), ptr(d[5]), d[4].size(0) if d[4] is not None else 0)
    out = (L, w_arr, cols, prep, dec, decoder[4].size(0) if decoder is not None else widths[-1])
    if cache is not None:
        cache[x_width] = out
    return out


@on_device_of
def static_infer_fwd(x, edge_attr, edge_index, plan_parts, layers, decoder, prepared=None, hint=PLAN_HINT_REFERENCE, gemm_mode=None, fuse_decoder=True, cache=None):
    """-> (logits | last activations, plan parts) or None when a shape is outside the fused kernels (nothing launched).
    `plan_parts` = (rowptr, src, eid) of an existing plan, or None: the plan is built inside the call from `edge_index` (int64 [2,E], any strides)
    and its arrays are returned.  `layers`: per conv layer (We, be, Wj, bj, Wi, scale | None, shift | None); `decoder` = (W0, b0, scale1 | None,
    shift1 | None, W3, b3) or None; `prepared`: per-layer dgnn_sage_layer_prepare buffers or None; `fuse_decoder`: the last layer's launch carries the
    decoder (prepared[-1] is then the block made WITH the decoder), else layer and decoder run apart.  `cache`: see _infer_tables."""
    _req(x, "x", dim=2)
    _req(edge_attr, "edge_attr", dim=2)
    n, dev = x.size(0), x.device
    L, w_arr, cols, prep, dec, n_out = _infer_tables(x.size(1), layers, decoder, prepared, cache)
    E = edge_index.size(1) if edge_index is not None else plan_parts[1].numel()
    build = plan_parts is None
    if build:
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        eid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        scratch = torch.empty(int(lib().dgnn_plan_scratch_elems(E, n)), dtype=torch.int32, device=dev)
        plan_parts = (rowptr, src, eid)
    else:
        rowptr, src, eid = plan_parts
        scratch = None
    logits = torch.empty((n, n_out), dtype=torch.float32, device=dev)
    work = torch.empty(int(lib().dgnn_static_infer_workspace_bytes(n, L, w_arr)), dtype=torch.uint8, device=dev)   # (the caching allocator hands out 512-byte aligned blocks)
    rc = lib().dgnn_static_infer_fwd(
        ptr(edge_index) if build else None, edge_index.stride(0) if build else 0, edge_index.stride(1) if build else 0, E, hint, ptr(rowptr), ptr(src),
        ptr(eid), ptr(scratch), n, ptr(x), _ld(x), ptr(edge_attr), _ld(edge_attr), edge_attr.size(1), L, w_arr, *cols, prep, *dec, int(bool(fuse_decoder)),
        GEMM_MODE if gemm_mode is None else gemm_mode, ptr(work), ptr(logits), stream_ptr())
    if rc == DGNN_E_UNSUPPORTED:
        return None
    check(rc, "dgnn_static_infer_fwd", poll=build)
    return logits, plan_parts


@on_device_of
def static_infer_rings_fwd(x, edge_attr, edge_index, plan_parts, n_dst, layers, decoder, prepared=None, hint=PLAN_HINT_GROUPED, gemm_mode=None, fuse_decoder=True,
                           attr_in_plan_order=True, cache=None):
    """One rank's part of a partitioned scene, rings of halo cells recomputed instead of exchanged (dgnn_static_infer_rings_fwd; arguments as
    static_infer_fwd): x [n_loc, F] local rows (owned cells, ring 1, ring 2, ...), `edge_index` the local list (in-edges of the first n_dst[0] cells),
    n_dst[l] = destinations of layer l.  -> (logits [n_dst[-1], n_logits], plan parts) or None (nothing launched)."""
    import ctypes as C
    _req(x, "x", dim=2)
    _req(edge_attr, "edge_attr", dim=2)
    n_loc, dev = x.size(0), x.device
    L, w_arr, cols, prep, dec, n_out = _infer_tables(x.size(1), layers, decoder, prepared, cache)
    assert len(n_dst) == L
    nd_arr = (C.c_int64 * L)(*[int(v) for v in n_dst])
    E = edge_index.size(1) if edge_index is not None else plan_parts[1].numel()
    build = plan_parts is None
    if build:
        rowptr = torch.empty(n_dst[0] + 1, dtype=torch.int32, device=dev)
        src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        eid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        scratch = torch.empty(int(lib().dgnn_plan_scratch_elems(E, n_dst[0])), dtype=torch.int32, device=dev)
        plan_parts = (rowptr, src, eid)
    else:
        rowptr, src, eid = plan_parts
        scratch = None
    logits = torch.empty((n_dst[-1], n_out), dtype=torch.float32, device=dev)
    work = torch.empty(int(lib().dgnn_static_infer_workspace_bytes(n_dst[0], L, w_arr)), dtype=torch.uint8, device=dev)
    rc = lib().dgnn_static_infer_rings_fwd(
        ptr(edge_index) if build else None, edge_index.stride(0) if build else 0, edge_index.stride(1) if build else 0, E, hint, ptr(rowptr), ptr(src),
        ptr(eid), ptr(scratch), int(bool(attr_in_plan_order)), n_loc, nd_arr, ptr(x), _ld(x), ptr(edge_attr), _ld(edge_attr), edge_attr.size(1),
        L, w_arr, *cols, prep, *dec, int(bool(fuse_decoder)), GEMM_MODE if gemm_mode is None else gemm_mode, ptr(work), ptr(logits), stream_ptr())
    if rc == DGNN_E_UNSUPPORTED:
        return None
    check(rc, "dgnn_static_infer_rings_fwd", poll=build)
    return logits, plan_parts


@on_device_of
def static_infer_rings_fwd_bf16(x, edge_attr, edge_index, plan_parts, n_dst, layers, decoder, hint=PLAN_HINT_GROUPED, attr_in_plan_order=True, cache=None):
    """static_infer_rings_fwd in bf16 STORAGE (dgnn_static_infer_rings_fwd_bf16): fp32 input rows read in place by the first layer, 16-bit rows (unsigned
    with BF16_UNSIGNED_ROWS) between the layers, the decoder inside the last launch.  A whole scene is n_dst = [n] * L.  None: nothing launched."""
    import ctypes as C
    _req(x, "x", dim=2)
    _req(edge_attr, "edge_attr", dim=2)
    n_loc, dev = x.size(0), x.device
    L, w_arr, cols, _, dec, n_out = _infer_tables(x.size(1), layers, decoder, None, cache)
    assert len(n_dst) == L and decoder is not None
    nd_arr = (C.c_int64 * L)(*[int(v) for v in n_dst])
    E = edge_index.size(1) if edge_index is not None else plan_parts[1].numel()
    build = plan_parts is None
    if build:
        rowptr = torch.empty(n_dst[0] + 1, dtype=torch.int32, device=dev)
        src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        eid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        scratch = torch.empty(int(lib().dgnn_plan_scratch_elems(E, n_dst[0])), dtype=torch.int32, device=dev)
        plan_parts = (rowptr, src, eid)
    else:
        rowptr, src, eid = plan_parts
        scratch = None
    logits = torch.empty((n_dst[-1], n_out), dtype=torch.float32, device=dev)
    work = torch.empty(int(lib().dgnn_static_infer_workspace_bytes(n_dst[0], L, w_arr)), dtype=torch.uint8, device=dev)
    mode = BF16_MODE | (BF16_ROWS_OUT_UNSIGNED if BF16_UNSIGNED_ROWS else 0)
    rc = lib().dgnn_static_infer_rings_fwd_bf16(
        ptr(edge_index) if build else None, edge_index.stride(0) if build else 0, edge_index.stride(1) if build else 0, E, hint, ptr(rowptr), ptr(src),
        ptr(eid), ptr(scratch), int(bool(attr_in_plan_order)), n_loc, nd_arr, ptr(x), _ld(x), ptr(edge_attr), _ld(edge_attr), edge_attr.size(1),
        L, w_arr, *cols, *dec, mode, ptr(work), ptr(logits), stream_ptr())
    if rc == DGNN_E_UNSUPPORTED:
        return None
    check(rc, "dgnn_static_infer_rings_fwd_bf16", poll=build)
    return logits, plan_parts


@on_device_of
def static_infer_partitioned_fwd(x, edge_attr, edge_index, plan_parts, n_own, n_interior, layers, decoder, prepared=None, halo=None, comm=None, send_buf=None,
                                 hint=PLAN_HINT_GROUPED, gemm_mode=None, fuse_decoder=True, attr_in_plan_order=True, cache=None):
    """One rank's part of a partitioned scene in one library call (dgnn_static_infer_partitioned_fwd; arguments as static_infer_fwd): x [n_own + n_halo, F]
    local rows, `edge_index` the local list (destinations < n_own), `halo` / `comm` the library's halo plan and communicator (None: a single-rank
    part), `send_buf` a uint8 buffer of n_send * max hidden width * 4 bytes.  -> (logits [n_own, n_logits], plan parts) or None (nothing launched)."""
    _req(x, "x", dim=2)
    _req(edge_attr, "edge_attr", dim=2)
    n_loc, dev = x.size(0), x.device
    n_halo = n_loc - n_own
    L, w_arr, cols, prep, dec, n_out = _infer_tables(x.size(1), layers, decoder, prepared, cache)
    E = edge_index.size(1) if edge_index is not None else plan_parts[1].numel()
    build = plan_parts is None
    if build:
        rowptr = torch.empty(n_own + 1, dtype=torch.int32, device=dev)
        src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        eid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E]
        scratch = torch.empty(int(lib().dgnn_plan_scratch_elems(E, n_own)), dtype=torch.int32, device=dev)
        plan_parts = (rowptr, src, eid)
    else:
        rowptr, src, eid = plan_parts
        scratch = None
    logits = torch.empty((n_own, n_out), dtype=torch.float32, device=dev)
    work = torch.empty(int(lib().dgnn_static_infer_workspace_bytes(n_loc, L, w_arr)), dtype=torch.uint8, device=dev)
    rc = lib().dgnn_static_infer_partitioned_fwd(
        ptr(edge_index) if build else None, edge_index.stride(0) if build else 0, edge_index.stride(1) if build else 0, E, hint, ptr(rowptr), ptr(src),
        ptr(eid), ptr(scratch), int(bool(attr_in_plan_order)), n_own, n_interior, n_halo, ptr(x), _ld(x), ptr(edge_attr), _ld(edge_attr), edge_attr.size(1),
        L, w_arr, *cols, prep, *dec, int(bool(fuse_decoder)), GEMM_MODE if gemm_mode is None else gemm_mode, halo, comm, ptr(send_buf), ptr(work), ptr(logits),
        stream_ptr())
    if rc == DGNN_E_UNSUPPORTED:
        return None
    check(rc, "dgnn_static_infer_partitioned_fwd", poll=build)
    return logits, plan_parts


def decoder_fused_supported(k: int, hidden: int, n_out: int) -> bool:
    return k == 128 and hidden == 64 and n_out in (1, 2)


@on_device_of
def decoder_fused_fwd(y, W0, b0, scale, shift, W3, b3):
    _req(y, "y", dim=2)
    M, n_out = y.size(0), W3.size(0)
    out = torch.empty((M, n_out), dtype=torch.float32, device=y.device)
    check(lib().dgnn_decoder_fused_fwd(ptr(y), _ld(y), M, y.size(1), ptr(W0.contiguous()), ptr(b0), ptr(scale), ptr(shift),
                                       W0.size(0), ptr(W3.contiguous()), ptr(b3), n_out, ptr(out), n_out, stream_ptr()),
          "dgnn_decoder_fused_fwd")
    return out


# ---- bf16 storage path (BASELINE config 3) ----------------------------------------------------------------------------------
BF16 = torch.bfloat16
# UNSIGNED rows (round 4; include/dgnn_hip.h DGNN_BF16_ROWS_*_UNSIGNED): rows written behind a ReLU keep the fp32 bits [30:15] -- 8 exponent + 8 mantissa
# bits, no sign -- in their 16 bits: half the storage rounding of bf16 in the same bytes.  Such rows travel in torch.int16 tensors (the dtype IS the
# format: it survives slicing, cat and the halo exchange), written and read by the fused bf16-storage layers only; rows_unsigned_to_bf16 converts for
# every other consumer.  DGNN_BF16_UNSIGNED_ROWS=0: plain bf16 rows everywhere (rounds 2-3).
UROWS = torch.int16
BF16_UNSIGNED_ROWS = __import__("os").environ.get("DGNN_BF16_UNSIGNED_ROWS", "1") != "0"
BF16_ROWS_IN_UNSIGNED, BF16_ROWS_OUT_UNSIGNED = 16, 32
BF16_SINGLE, BF16_COMPENSATED = 0, 1
# how the fused bf16-storage kernels feed the matrix cores: "single" = every operand rounded to bf16 once; "compensated"
# (default) = only the stored activations are bf16, the fp32 mean / attributes / parameters go in as (hi, lo) bf16 pairs
BF16_MODE = {"single": BF16_SINGLE, "compensated": BF16_COMPENSATED}[__import__("os").environ.get("DGNN_BF16_MODE", "compensated")]


@on_device_of
def cast_to_bf16(x: torch.Tensor, cols_pad: int = None) -> torch.Tensor:
    """fp32 [n, cols] (any row stride) -> bf16 [n, cols_pad] with zero padding columns (cols_pad even, default: cols rounded up to even)."""
    _req(x, "x", dim=2)
    n, cols = x.shape
    if cols_pad is None:
        cols_pad = (cols + 1) // 2 * 2
    out = torch.empty((n, cols_pad), dtype=BF16, device=x.device)
    check(lib().dgnn_cast_f32_to_bf16(ptr(x), _ld(x), n, cols, cols_pad, ptr(out), cols_pad, stream_ptr()), "dgnn_cast_f32_to_bf16")
    return out


@on_device_of
def rows_unsigned_to_bf16(x: torch.Tensor) -> torch.Tensor:
    """unsigned rows (int16 container) -> plain bf16 rows, round to nearest even"""
    _req(x, "x", UROWS, dim=2)
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib().dgnn_rows_unsigned_to_bf16(ptr(x), _ld(x), x.size(0), x.size(1), ptr(out), x.size(1), stream_ptr()), "dgnn_rows_unsigned_to_bf16")
    return out


@on_device_of
def cast_to_f32(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == UROWS:
        x = rows_unsigned_to_bf16(x)
    _req(x, "x", BF16, dim=2)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().dgnn_cast_bf16_to_f32(ptr(x), _ld(x), x.size(0), x.size(1), ptr(out), x.size(1), stream_ptr()), "dgnn_cast_bf16_to_f32")
    return out


def fused_layer_supported_bf16(c_in: int, c_out: int, f_e: int, x: torch.Tensor = None) -> bool:
    """Mirrors the shape dispatch of dgnn_sage_layer_fused_fwd_bf16 (csrc/fused_bf16.hip).  `x` fp32: the first layer reading the
    caller's fp32 features in place (c_in <= 32)."""
    if not FUSED_ENABLED or f_e != 20 or c_in > 128 or c_out not in (64, 128):
        return False
    nb = 2 if c_in <= 32 else (4 if c_in <= 64 else 8)
    ok = c_in % nb == 0
    if x is not None and x.dtype == torch.float32:
        return ok and c_in <= 32 and x.data_ptr() % 4 == 0
    if x is not None:
        ok = ok and x.stride(0) % nb == 0 and x.data_ptr() % (2 * nb) == 0
    return ok


@on_device_of
def sage_layer_fused_fwd_bf16(rowptr, src, n_dst, x_src, c_in, edge_attr, We, be, Wj, bj, Wi, scale, shift, relu, out=None, eid=None,
                              x_dst=None, rows_out_unsigned=False):
    """bf16-storage twin of sage_layer_fused_fwd.  `x_src` bf16 [n_src, ld >= c_in] (padding columns allowed: `c_in` is the
    logical width = Wj.shape[1]) -- or, for the first layer (c_in <= 32), the caller's fp32 rows read in place; returns bf16
    [n_dst, c_out]."""
    _req(x_src, "x_src", ACT + (UROWS,), dim=2)
    _same(x_dst, x_src, "x_dst")
    c_out = Wj.size(0)
    if out is None:
        out = torch.empty((n_dst, c_out), dtype=UROWS if rows_out_unsigned else BF16, device=x_src.device)
    else:
        _req(out, "out", (BF16, UROWS), dim=2)      # the buffer's dtype names the row format it receives
        if out.size(0) < n_dst or out.size(1) != c_out or out.stride(0) != c_out:
            raise ValueError("out must be a contiguous [>= n_dst, c_out] buffer")
    fmt = (BF16_ROWS_IN_UNSIGNED if x_src.dtype == UROWS else 0) | (BF16_ROWS_OUT_UNSIGNED if out.dtype == UROWS else 0)
    check(lib().dgnn_sage_layer_fused_fwd_bf16(
        ptr(rowptr), ptr(src), ptr(eid), n_dst, ptr(x_src), int(x_src.dtype == torch.float32), ptr(x_dst), _ld(x_src), c_in, ptr(edge_attr),
        _ld(edge_attr), We.size(1),
        ptr(We), ptr(be), ptr(Wj), ptr(bj), ptr(Wi), ptr(scale), ptr(shift), int(bool(relu)), c_out, ptr(out), c_out, BF16_MODE | fmt, stream_ptr()),
        "dgnn_sage_layer_fused_fwd_bf16")
    return out


def fused_layer_decoder_supported_bf16(c_in: int, c_out: int, f_e: int, hidden: int, n_out: int, x=None) -> bool:
    """the last bf16-storage layer's launch can carry the decoder: shipped shape, compensated arithmetic, bf16 rows 16-byte aligned"""
    ok = (FUSE_DECODER and FUSED_ENABLED and BF16_MODE == BF16_COMPENSATED and 64 < c_in <= 128 and c_in % 8 == 0 and c_out == 128 and f_e == 20
          and hidden == 64 and n_out == 2)
    if ok and x is not None:
        ok = x.dtype in (torch.bfloat16, UROWS) and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
    return ok


@on_device_of
def sage_layer_fused_decoder_fwd_bf16(rowptr, src, n_dst, x_src, c_in, edge_attr, We, be, Wj, bj, Wi, scale, shift, relu, W0, b0, scale1, shift1, W3, b3,
                                      out=None, eid=None, x_dst=None):
    """Last conv layer (bf16 storage) + decoder, one launch -> fp32 logits [n_dst, 2]; the layer's output is never rounded to bf16."""
    _req(x_src, "x_src", (BF16, UROWS), dim=2)
    _same(x_dst, x_src, "x_dst")
    if out is None:
        out = torch.empty((n_dst, 2), dtype=torch.float32, device=x_src.device)
    else:
        _req(out, "out", dim=2)
        if out.size(0) < n_dst or out.size(1) != 2 or out.stride(0) != 2:
            raise ValueError("out must be a contiguous [>= n_dst, 2] buffer")
    check(lib().dgnn_sage_layer_fused_decoder_fwd_bf16(
        ptr(rowptr), ptr(src), ptr(eid), n_dst, ptr(x_src), ptr(x_dst), _ld(x_src), c_in, ptr(edge_attr), _ld(edge_attr), We.size(1), ptr(We), ptr(be),
        ptr(Wj), ptr(bj), ptr(Wi), ptr(scale), ptr(shift), int(bool(relu)), Wj.size(0), ptr(W0.contiguous()), ptr(b0), ptr(scale1), ptr(shift1), W0.size(0),
        ptr(W3.contiguous()), ptr(b3), W3.size(0), ptr(out), BF16_MODE | (BF16_ROWS_IN_UNSIGNED if x_src.dtype == UROWS else 0), stream_ptr()),
        "dgnn_sage_layer_fused_decoder_fwd_bf16")
    return out


@on_device_of
def decoder_fused_fwd_bf16(y, W0, b0, scale, shift, W3, b3):
    _req(y, "y", BF16, dim=2)
    M, n_out = y.size(0), W3.size(0)
    out = torch.empty((M, n_out), dtype=torch.float32, device=y.device)
    check(lib().dgnn_decoder_fused_fwd_bf16(ptr(y), _ld(y), M, y.size(1), ptr(W0.contiguous()), ptr(b0), ptr(scale), ptr(shift),
                                            W0.size(0), ptr(W3.contiguous()), ptr(b3), n_out, ptr(out), n_out, BF16_MODE, stream_ptr()),
          "dgnn_decoder_fused_fwd_bf16")
    return out


# ---- training-mode conv layer, one call each way (csrc/train.hip) -----------------------------------------------------------
TRAIN_COMPOSITE = __import__("os").environ.get("DGNN_TRAIN_COMPOSITE", "1") != "0"
TRAIN_WHOLE_MODEL = __import__("os").environ.get("DGNN_TRAIN_WHOLE_MODEL", "1") != "0"   # Static fp32: all layers in one call each way
TRAIN_DECODER_OUTPUT_IN_CALL = __import__("os").environ.get("DGNN_TRAIN_DECODER_IN_CALL", "1") != "0"   # ... the decoder's output Linear too
# the direct training step keeps ONE gradient buffer per model and rewrites it in place (p.grad stays the same tensor from step to step, as after
# zero_grad(set_to_none=False)); 0 = fresh gradient tensors every step
TRAIN_KEEP_GRADS = __import__("os").environ.get("DGNN_TRAIN_KEEP_GRADS", "1") != "0"


@on_device_of
def sage_layer_train_fwd(plan_parts, n_dst, x, edge_attr, We, be, Wj, bj, Wi, gamma, beta, running_mean, running_var, momentum, eps, relu):
    """conv (aggregate + lin_j + lin_i) -> BatchNorm(batch statistics, running buffers updated) -> ReLU as ONE library call; fp32 or
    bf16 storage (the type of x).  `plan_parts` = (rowptr, src, eid) of the destination-sorted plan, or None for a plain
    Linear + BatchNorm block (decoder).  -> (y, a | None, z, stats[4, c_out] = mean, var, scale, shift)"""
    _req(x, "x", ACT, dim=2)
    c_in, c_out = x.size(1), Wj.size(0)
    dev, dt = x.device, x.dtype
    rowptr = src = eid = a = None
    f_e = 0
    if plan_parts is not None:
        rowptr, src, eid = plan_parts
        a = torch.empty((n_dst, c_in), dtype=dt, device=dev)
        if We is not None:
            _req(edge_attr, "edge_attr", dim=2)
            f_e = We.size(1)
            if We.size(0) != c_in or edge_attr.size(1) < f_e:
                raise ValueError("lin_e weight %s does not match c_in=%d / edge_attr %s" % (tuple(We.shape), c_in, tuple(edge_attr.shape)))
    if Wj.size(1) != c_in or (Wi is not None and tuple(Wi.shape) != tuple(Wj.shape)):
        raise ValueError("lin_j / lin_i weights do not match c_in=%d" % c_in)
    z = torch.empty((n_dst, c_out), dtype=dt, device=dev)
    y = torch.empty((n_dst, c_out), dtype=dt, device=dev)
    stats = torch.empty((4, c_out), dtype=torch.float32, device=dev)
    scratch = _f32(lib().dgnn_colstats_scratch_elems(n_dst, c_out), dev)
    head = (ptr(rowptr), ptr(src), ptr(eid), n_dst, ptr(x), _ld(x), c_in, ptr(edge_attr) if f_e else None, _ld(edge_attr) if f_e else 0, f_e,
            ptr(We), ptr(be), ptr(Wj), ptr(bj), ptr(Wi), c_out, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(momentum), float(eps),
            int(bool(relu)), ptr(a), ptr(z), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), ptr(y), ptr(scratch))
    if dt == torch.bfloat16:
        check(lib().dgnn_sage_layer_train_fwd_bf16(*head, stream_ptr()), "dgnn_sage_layer_train_fwd_bf16")
    else:
        check(lib().dgnn_sage_layer_train_fwd(*head, GEMM_MODE, stream_ptr()), "dgnn_sage_layer_train_fwd")
    _written_behind_torch(running_mean, running_var)
    return y, a, z, stats


@on_device_of
def sage_layer_train_bwd(t_parts, rowptr_dst, n_src, n_dst, x, edge_attr, We, be, Wj, Wi, has_bias, gamma, stats, eps, relu, a, z, y, dy, need_dx):
    """backward of sage_layer_train_fwd as ONE library call -> (dx | None, dWe, dbe, dWj, dbj, dWi, dgamma, dbeta); parameter
    gradients are fp32 views of one buffer, dx has the storage type of x."""
    c_in, c_out = x.size(1), Wj.size(0)
    dev, dt = x.device, x.dtype
    agg = t_parts is not None
    f_e = We.size(1) if (agg and We is not None) else 0
    sizes = [c_in * f_e, c_in if f_e else 0, c_out * c_in, c_out if has_bias else 0, c_out * c_in if (agg and Wi is not None) else 0, c_out, c_out]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    shapes = [(c_in, f_e), (c_in,), (c_out, c_in), (c_out,), (c_out, c_in), (c_out,), (c_out,)]
    parts, o = [], 0
    for s, sh in zip(sizes, shapes):       # one view op per gradient, already in its final shape
        parts.append(torch.as_strided(flat, sh, (sh[1], 1) if len(sh) == 2 else (1,), o) if s else None)
        o += s
    dWe, dbe, dWj, dbj, dWi, dgamma, dbeta = parts
    dx = torch.empty((n_src if agg else n_dst, c_in), dtype=dt, device=dev) if need_dx else None
    scratch = _f32(lib().dgnn_sage_layer_train_scratch_elems(n_src, n_dst, c_in, c_out, f_e), dev)
    t_rowptr, t_dst, t_eid = t_parts if agg else (None, None, None)
    head = (ptr(t_rowptr), ptr(t_dst), ptr(t_eid), ptr(rowptr_dst), n_src, n_dst, ptr(x), _ld(x), c_in, ptr(edge_attr) if f_e else None,
            _ld(edge_attr) if f_e else 0, f_e, ptr(We), ptr(be), ptr(Wj), ptr(Wi), c_out, ptr(gamma), ptr(stats[0]), ptr(stats[1]), float(eps),
            int(bool(relu)), ptr(a), ptr(z), ptr(y), ptr(dy), ptr(dx), ptr(dWe), ptr(dbe), ptr(dWj), ptr(dbj), ptr(dWi), ptr(dgamma), ptr(dbeta))
    if dt == torch.bfloat16:
        dz = torch.empty((n_dst, c_out), dtype=dt, device=dev)
        da = torch.empty((n_dst, c_in), dtype=dt, device=dev) if agg else None
        check(lib().dgnn_sage_layer_train_bwd_bf16(*head, ptr(dz), ptr(da), ptr(scratch), stream_ptr()), "dgnn_sage_layer_train_bwd_bf16")
    else:
        check(lib().dgnn_sage_layer_train_bwd(*head, ptr(scratch), GEMM_MODE, stream_ptr()), "dgnn_sage_layer_train_bwd")
    return (dx, dWe, dbe, dWj, dbj, dWi, dgamma, dbeta)


# ---- all conv layers of the Updated variant per call (csrc/train.hip: dgnn_updated_stack_fwd / _bwd) ---------------------------------------
UPDATED_STACK = __import__("os").environ.get("DGNN_UPDATED_STACK", "1") != "0"
_UPDATED_ARRAYS = {}
UPDATED_TAIL_IN_CALL = __import__("os").environ.get("DGNN_UPDATED_TAIL_IN_CALL", "1") != "0"   # ... and the model's output network behind it


@on_device_of
def updated_stack_fwd(x0, edge_attr_all, pos, layers):
    """`layers`: per conv layer a dict with plan (GraphPlan), e_id (int64 [E_l], rows of the scene's edge tensor), rows0 (int32 form of e_id, layer 0),
    edge_in, relu, We, be, Wl, bl, Wr.  -> (y_last, saved): every layer's ea / phi / a / y (+ the chaining's inverse maps) in one buffer."""
    import ctypes as C
    _req(x0, "x", ACT, dim=2)
    _req(edge_attr_all, "edge_attr", dim=2)
    dev, dt, L = x0.device, x0.dtype, len(layers)
    bf = dt == torch.bfloat16
    widths = [x0.size(1)] + [l["Wl"].size(0) for l in layers]
    esz = 2 if bf else 4
    offs, off = [], 0          # byte offsets into one buffer, every piece 16-byte aligned

    def take(nbytes):
        nonlocal off
        o = off
        off += (nbytes + 15) & ~15
        return o
    for i, l in enumerate(layers):
        p = l["plan"]
        E, n, ci, co, k = p.E, p.n_dst, widths[i], widths[i + 1], l["edge_in"]
        if l["Wl"].size(1) != ci or l["We"].size(0) != ci or l["We"].size(1) != k:
            raise ValueError("Updated conv %d: lin_e %s / lin_l %s do not match %d input channels, %d edge columns" % (i, tuple(l["We"].shape), tuple(l["Wl"].shape), ci, k))
        ld_ea = (k + 1) // 2 * 2 if (bf and i == 0) else k
        offs.append(dict(ea=take(max(E, 1) * ld_ea * esz), ld_ea=ld_ea, phi=take(max(E, 1) * ci * esz), a=take(n * ci * esz), y=take(n * co * esz),
                         inv=take(4 * max(layers[i - 1]["plan"].E, 1)) if i else None))
    ea0 = take(4 * max(layers[0]["plan"].E, 1) * layers[0]["edge_in"]) if bf else None
    buf = torch.empty(off, dtype=torch.uint8, device=dev)
    base = buf.data_ptr()
    at = lambda o: None if o is None else base + o
    parts = [l["plan"].part_ptrs(False) for l in layers]
    arr = lambda k: _parr([l[k] for l in layers])
    i64 = lambda v: _iarr(v, C.c_int64)
    i32 = lambda v: _iarr(v, C.c_int32)
    # tables of what belongs to the MODEL (parameter addresses, widths, edge_in, relu flags): rebuilt only when an address has moved
    names = ("We", "be", "Wl", "bl", "Wr")
    key = tuple((l[k].data_ptr() if l[k] is not None else 0) for l in layers for k in names) + tuple((l["edge_in"], bool(l["relu"])) for l in layers) + (x0.size(1),)
    hit = _UPDATED_ARRAYS.get(L)
    if hit is None or hit[0] != key:
        hit = (key, dict({k: arr(k) for k in names}, widths=i32(widths), edge_in=i32([l["edge_in"] for l in layers]),
                         relu=i32([int(bool(l["relu"])) for l in layers])))
        _UPDATED_ARRAYS[L] = hit
    pa = hit[1]
    check(lib().dgnn_updated_stack_fwd(
        L, _parr([p[0] for p in parts]), _parr([p[1] for p in parts]), _parr([p[2] for p in parts]), arr("e_id"), ptr(layers[0]["rows0"]),
        i64([l["plan"].n_dst for l in layers]), i64([l["plan"].E for l in layers]), ptr(x0), _ld(x0), pa["widths"], pa["edge_in"],
        ptr(edge_attr_all), _ld(edge_attr_all), edge_attr_all.size(0), ptr(pos), pa["We"], pa["be"], pa["Wl"], pa["bl"], pa["Wr"],
        pa["relu"], _parr([at(o["ea"]) for o in offs]), i64([o["ld_ea"] for o in offs]), at(ea0),
        _parr([at(o["phi"]) for o in offs]), _parr([at(o["a"]) for o in offs]), _parr([at(o["y"]) for o in offs]), _parr([at(o["inv"]) for o in offs]),
        int(bf), GEMM_MODE, stream_ptr()), "dgnn_updated_stack_fwd", poll=True)
    n, co = layers[-1]["plan"].n_dst, widths[-1]
    y = torch.as_strided(buf.view(dt), (n, co), (co, 1), offs[-1]["y"] // esz)
    return y, (buf, offs, widths, pa)


@on_device_of
def updated_stack_bwd(x0, layers, saved, dy):
    """-> per layer (dWe, dbe, dWl, dbl | None, dWr | None): views of one fp32 buffer"""
    import ctypes as C
    buf, offs, widths, pa = saved
    dev, dt, L = x0.device, x0.dtype, len(layers)
    bf = dt == torch.bfloat16
    base = buf.data_ptr()
    at = lambda o: None if o is None else base + o
    sizes, off = [], 0
    for i, l in enumerate(layers):
        ci, co, k = widths[i], widths[i + 1], l["edge_in"]
        row = []
        for sz in (ci * k, ci, co * ci, co if l["bl"] is not None else 0, co * ci if l["Wr"] is not None else 0):
            row.append((off, sz) if sz else None)
            off += sz
        sizes.append(row)
    flat = torch.empty(off, dtype=torch.float32, device=dev)
    gbase = flat.data_ptr()
    gat = lambda e: None if e is None else gbase + 4 * e[0]
    P = [l["plan"] for l in layers]
    mx = lambda vals: max(list(vals) + [1])
    # the seven work buffers of the backward (storage type) as pieces of one allocation, 16-byte aligned
    esz = 2 if bf else 4
    counts = [mx(P[i].n_src * widths[i] for i in range(1, L)) if L > 1 else 0] * 2 + [
        mx(P[i].E * layers[i]["edge_in"] for i in range(1, L)) if L > 1 else 0, mx(P[i].E * widths[i] for i in range(L - 1)) if L > 1 else 0,
        mx(P[i].n_dst * widths[i + 1] for i in range(L)), mx(2 * P[i].n_dst * widths[i] for i in range(L)), mx(P[i].E * widths[i] for i in range(L))]
    woffs, wtot = [], 0
    for c in counts:
        woffs.append(wtot)
        wtot += (c * esz + 15) & ~15
    wbuf = torch.empty(max(wtot, 16), dtype=torch.uint8, device=dev)
    wbase = wbuf.data_ptr()
    wptr = [(wbase + o) if c else None for o, c in zip(woffs, counts)]
    dxb, d_ea, dphi_ext, dz, da, dphi = wptr[0:2], wptr[2], wptr[3], wptr[4], wptr[5], wptr[6]
    scratch = _f32(max(lib().dgnn_sage_updated_train_scratch_elems(P[i].n_dst, P[i].E, widths[i], widths[i + 1], layers[i]["edge_in"]) for i in range(L)), dev)
    tps = [p.transposed_ptrs(False) for p in P]
    arr = lambda k: _parr([l[k] for l in layers])
    i64 = lambda v: _iarr(v, C.c_int64)
    i32 = lambda v: _iarr(v, C.c_int32)
    col = lambda j: _parr([gat(r[j]) for r in sizes])
    check(lib().dgnn_updated_stack_bwd(
        L, _parr([t[0] for t in tps]), _parr([t[1] for t in tps]), _parr([t[2] for t in tps]), _parr([p.part_ptrs(False)[0] for p in P]),
        i64([p.n_src for p in P]), i64([p.n_dst for p in P]), i64([p.E for p in P]), ptr(x0), _ld(x0), pa["widths"], pa["edge_in"],
        pa["We"], pa["Wl"], pa["Wr"], pa["relu"], _parr([at(o["ea"]) for o in offs]), i64([o["ld_ea"] for o in offs]),
        _parr([at(o["phi"]) for o in offs]), _parr([at(o["a"]) for o in offs]), _parr([at(o["y"]) for o in offs]), _parr([at(o["inv"]) for o in offs]),
        ptr(dy), col(0), col(1), col(2), col(3), col(4), _parr(dxb), d_ea, dphi_ext, dz, da, dphi, ptr(scratch), int(bf), GEMM_MODE,
        stream_ptr()), "dgnn_updated_stack_bwd")
    st = torch.as_strided
    grads = []
    for i, (l, row) in enumerate(zip(layers, sizes)):
        ci, co, k = widths[i], widths[i + 1], l["edge_in"]
        m = lambda e, r, c: None if e is None else st(flat, (r, c), (c, 1), e[0])
        v = lambda e, n: None if e is None else st(flat, (n,), (1,), e[0])
        grads.append((m(row[0], ci, k), v(row[1], ci), m(row[2], co, ci), v(row[3], co), m(row[4], co, ci)))
    return grads


@on_device_of
def updated_tail_fwd(x, W1, b1, W3, b3):
    """out_net behind the conv stack: -> (logits fp32 [n, n_out], h [n, hdim] in x's type)"""
    n, c, hd, no = x.size(0), x.size(1), W1.size(0), W3.size(0)
    h = torch.empty((n, hd), dtype=x.dtype, device=x.device)
    logits = torch.empty((n, no), dtype=torch.float32, device=x.device)
    check(lib().dgnn_updated_tail_fwd(n, ptr(x), _ld(x), c, ptr(W1), ptr(b1), hd, ptr(W3), ptr(b3), no, ptr(h), ptr(logits), int(x.dtype == torch.bfloat16),
                                      GEMM_MODE, stream_ptr()), "dgnn_updated_tail_fwd")
    return logits, h


@on_device_of
def updated_tail_bwd(x, W1, W3, h, g):
    """-> (dx [n, c] in x's type, dW1, db1, dW3, db3)"""
    n, c, hd, no = x.size(0), x.size(1), W1.size(0), W3.size(0)
    flat = torch.empty(hd * c + hd + no * hd + no, dtype=torch.float32, device=x.device)
    st = torch.as_strided
    dW1, db1 = st(flat, (hd, c), (c, 1), 0), st(flat, (hd,), (1,), hd * c)
    dW3, db3 = st(flat, (no, hd), (hd, 1), hd * c + hd), st(flat, (no,), (1,), hd * c + hd + no * hd)
    dx = torch.empty((n, c), dtype=x.dtype, device=x.device)
    dh = torch.empty((n, hd), dtype=x.dtype, device=x.device)
    scratch = _f32(lib().dgnn_updated_tail_scratch_elems(n, c, hd, no), x.device)
    check(lib().dgnn_updated_tail_bwd(n, ptr(x), _ld(x), c, ptr(W1), hd, ptr(W3), no, ptr(h), ptr(g), ptr(dW1), ptr(db1), ptr(dW3), ptr(db3), ptr(dx), ptr(dh),
                                      ptr(scratch), int(x.dtype == torch.bfloat16), GEMM_MODE, stream_ptr()), "dgnn_updated_tail_bwd")
    return dx, dW1, db1, dW3, db3


# ---- edge-embedding chaining of the Updated variant (csrc/chain.hip) ---------------------------------------------------------
@on_device_of
def edge_chain_fwd(phi, e_id_cur, e_id_next, c, pos, relu):
    """-> (out [n_next, c] = relu?(zeros[E_all, C]; [e_id_cur] = phi)[e_id_next, :c], inv int32 [n_cur]); `pos`: int32 [E_all]
    table holding -1 everywhere (left that way)."""
    _req(phi, "phi", ACT, dim=2)
    _req(e_id_cur, "e_id_cur", torch.int64, 1)
    _req(e_id_next, "e_id_next", torch.int64, 1)
    _req(pos, "pos", torch.int32, 1)
    n_cur, n_next = e_id_cur.numel(), e_id_next.numel()
    if phi.size(0) != n_cur or not 0 < c <= phi.size(1):
        raise ValueError("phi %s does not match e_id_cur (%d) / c=%d" % (tuple(phi.shape), n_cur, c))
    e_id_cur, e_id_next = e_id_cur.contiguous(), e_id_next.contiguous()
    out = torch.empty((n_next, c), dtype=phi.dtype, device=phi.device)
    inv = torch.empty(max(n_cur, 1), dtype=torch.int32, device=phi.device)[:n_cur]
    check(getattr(lib(), "dgnn_edge_chain_fwd" + _sfx(phi))(ptr(phi), _ld(phi), int(c), ptr(e_id_cur), n_cur, ptr(e_id_next), n_next, pos.numel(),
                                                            ptr(pos), int(bool(relu)), ptr(out), c, ptr(inv), stream_ptr()),
          "dgnn_edge_chain_fwd", poll=True)
    return out, inv


@on_device_of
def edge_chain_bwd(g, phi, inv, c, relu):
    _req(phi, "phi", ACT, dim=2)
    _same(_req(g, "g", ACT, dim=2), phi, "g")
    dphi = torch.empty((phi.size(0), phi.size(1)), dtype=phi.dtype, device=phi.device)
    check(getattr(lib(), "dgnn_edge_chain_bwd" + _sfx(phi))(ptr(g), _ld(g), ptr(phi), _ld(phi), ptr(inv), phi.size(0), int(c), phi.size(1),
                                                            int(bool(relu)), ptr(dphi), phi.size(1), stream_ptr()), "dgnn_edge_chain_bwd")
    return dphi


# ---- volume-weighted KL cell loss (csrc/loss.hip) ------------------------------------------------------------------------------
CELL_NORMS = {None: 0, "": 0, "none": 0, "log": 1, "sqrt": 2}


@on_device_of
def kl_cell_loss_fwd(logits, gt, vol, norm: int):
    """-> (loss fp32 0-dim, sums fp64 [3] = sum cell*w, sum w, OA count)"""
    _req(logits, "logits", dim=2)
    _req(gt, "gt", dim=2)
    _req(vol, "vol", dim=1)
    n = logits.size(0)
    if logits.size(1) != 2 or gt.size(1) < 2 or gt.size(0) != n or vol.numel() != n:
        raise ValueError("kl_cell_loss: logits [n,2], gt [n,>=2], vol [n] expected, got %s %s %s" % (tuple(logits.shape), tuple(gt.shape), tuple(vol.shape)))
    out = torch.empty(4, dtype=torch.float64, device=logits.device)   # sums[3] | the fp32 loss in the first half of the 4th
    loss = out[3:].view(torch.float32)[:1].view(())
    scratch = torch.empty(int(lib().dgnn_kl_cell_loss_scratch_doubles(n)), dtype=torch.float64, device=logits.device)
    check(lib().dgnn_kl_cell_loss_fwd(ptr(logits), _ld(logits), ptr(gt), _ld(gt), ptr(vol), vol.stride(0), int(norm), n, ptr(out), ptr(loss), ptr(scratch),
                                      stream_ptr()), "dgnn_kl_cell_loss_fwd")
    return loss, out[:3]


@on_device_of
def kl_cell_loss_bwd(logits, gt, vol, norm: int, sums, grad_loss):
    dl = torch.empty((logits.size(0), 2), dtype=torch.float32, device=logits.device)
    grad_loss = grad_loss.to(torch.float32).contiguous()
    check(lib().dgnn_kl_cell_loss_bwd(ptr(logits), _ld(logits), ptr(gt), _ld(gt), ptr(vol), vol.stride(0), int(norm), logits.size(0), ptr(sums),
                                      ptr(grad_loss), ptr(dl), 2, stream_ptr()), "dgnn_kl_cell_loss_bwd")
    return dl


@on_device_of
def kl_cell_loss_step(logits, gt, vol, norm: int, running=None, grad_loss=None, backward=True):
    """kl_cell_loss_fwd + (`running` fp64 [3] += sums) + kl_cell_loss_bwd as ONE launch -> (loss, sums, dlogits | None); the same bits as the two
    calls.  None when the library declines the size (more than 65536 rows): the caller then makes the two calls."""
    _req(logits, "logits", dim=2)
    _req(gt, "gt", dim=2)
    _req(vol, "vol", dim=1)
    n = logits.size(0)
    if logits.size(1) != 2 or gt.size(1) < 2 or gt.size(0) != n or vol.numel() != n:
        raise ValueError("kl_cell_loss: logits [n,2], gt [n,>=2], vol [n] expected, got %s %s %s" % (tuple(logits.shape), tuple(gt.shape), tuple(vol.shape)))
    if running is not None and (running.dtype != torch.float64 or running.numel() < 3 or not running.is_contiguous() or running.device != logits.device):
        raise ValueError("kl_cell_loss_step: `running` must be a contiguous fp64 [3] tensor on the logits' device")
    out = torch.empty(4, dtype=torch.float64, device=logits.device)   # sums[3] | the fp32 loss in the first half of the 4th
    loss = out[3:].view(torch.float32)[:1].view(())
    dl = torch.empty((n, 2), dtype=torch.float32, device=logits.device) if backward else None
    rc = lib().dgnn_kl_cell_loss_step(ptr(logits), _ld(logits), ptr(gt), _ld(gt), ptr(vol), vol.stride(0), int(norm), n, ptr(grad_loss), ptr(out), ptr(loss),
                                      ptr(running), ptr(dl), 2, stream_ptr())
    if rc == DGNN_E_UNSUPPORTED:
        return None
    check(rc, "dgnn_kl_cell_loss_step")
    return loss, out[:3], dl


# ---- Updated variant: one conv layer per call each way (csrc/train.hip) --------------------------------------------------------
@on_device_of
def sage_updated_train_fwd(plan_parts, n_dst, x, ea, We, be, Wl, bl, Wr, relu):
    """-> (y [n_dst,c_out], phi [E,c_in], a [n_dst,c_in]) in the storage type of x (fp32 or bf16)"""
    _req(x, "x", ACT, dim=2)
    _same(_req(ea, "edge_attr", ACT, dim=2), x, "edge_attr")
    rowptr, src, eid = plan_parts
    c_in, c_out, k_e, E = x.size(1), Wl.size(0), We.size(1), ea.size(0)
    if We.size(0) != c_in or ea.size(1) != k_e or Wl.size(1) != c_in:
        raise ValueError("Updated conv: lin_e %s / lin_l %s do not match x %s, edge_attr %s" % (tuple(We.shape), tuple(Wl.shape), tuple(x.shape), tuple(ea.shape)))
    phi = torch.empty((E, c_in), dtype=x.dtype, device=x.device)
    a = torch.empty((n_dst, c_in), dtype=x.dtype, device=x.device)
    y = torch.empty((n_dst, c_out), dtype=x.dtype, device=x.device)
    check(lib().dgnn_sage_updated_train_fwd(ptr(rowptr), ptr(src), ptr(eid), n_dst, ptr(x), _ld(x), c_in, ptr(ea), _ld(ea), k_e, E, ptr(We), ptr(be), ptr(Wl),
                                            ptr(bl), ptr(Wr), c_out, int(bool(relu)), ptr(phi), ptr(a), ptr(y), int(x.dtype == torch.bfloat16), GEMM_MODE,
                                            stream_ptr()), "dgnn_sage_updated_train_fwd")
    return y, phi, a


@on_device_of
def sage_updated_train_bwd(t_parts, rowptr_dst, n_src, n_dst, x, ea, We, Wl, Wr, has_bias, relu, phi, a, y, dy, dphi_ext, need_dx, need_dea):
    """-> (dx | None, d_ea | None, dWe, dbe, dWl, dbl | None, dWr | None); parameter gradients are views of one fp32 buffer"""
    c_in, c_out, k_e, E = x.size(1), Wl.size(0), We.size(1), ea.size(0)
    dev, dt = x.device, x.dtype
    sizes = [c_in * k_e, c_in, c_out * c_in, c_out if has_bias else 0, c_out * c_in if Wr is not None else 0]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    shapes = [(c_in, k_e), (c_in,), (c_out, c_in), (c_out,), (c_out, c_in)]
    parts, o = [], 0
    for sz, sh in zip(sizes, shapes):      # one view op per gradient, already in its final shape
        parts.append(torch.as_strided(flat, sh, (sh[1], 1) if len(sh) == 2 else (1,), o) if sz else None)
        o += sz
    dWe, dbe, dWl, dbl, dWr = parts
    dx = torch.empty((n_src, c_in), dtype=dt, device=dev) if need_dx else None
    d_ea = torch.empty((E, k_e), dtype=dt, device=dev) if need_dea else None
    dz = torch.empty((n_dst, c_out), dtype=dt, device=dev) if relu else None
    da = torch.empty((n_dst, 2 * c_in), dtype=dt, device=dev)     # [da | dz . Wr] of the merged input-gradient GEMM
    dphi = torch.empty((max(E, 1), c_in), dtype=dt, device=dev)
    scratch = _f32(lib().dgnn_sage_updated_train_scratch_elems(n_dst, E, c_in, c_out, k_e), dev)
    t_rowptr, t_dst, t_eid = t_parts
    check(lib().dgnn_sage_updated_train_bwd(ptr(t_rowptr), ptr(t_dst), ptr(t_eid), ptr(rowptr_dst), n_src, n_dst, E, ptr(x), _ld(x), c_in, ptr(ea), _ld(ea), k_e,
                                            ptr(We), ptr(Wl), ptr(Wr), c_out, int(bool(relu)), ptr(phi), ptr(a), ptr(y), ptr(dy), ptr(dphi_ext), ptr(dx), ptr(d_ea),
                                            ptr(dWe), ptr(dbe), ptr(dWl), ptr(dbl), ptr(dWr), ptr(dz), ptr(da), ptr(dphi), ptr(scratch),
                                            int(dt == torch.bfloat16), GEMM_MODE, stream_ptr()), "dgnn_sage_updated_train_bwd")
    return (dx, d_ea, dWe, dbe, dWl, dbl, dWr)


# ---- Static model in training mode, all layers per call (csrc/train.hip) -------------------------------------------------------
def _parr(vals):
    import ctypes as C
    return (C.c_void_p * len(vals))(*[(v.data_ptr() if isinstance(v, torch.Tensor) else v) for v in vals])


def _iarr(vals, ty):
    return (ty * len(vals))(*vals)


_PARAM_ARRAYS = {}      # pointer tables of a model's parameters and BatchNorm buffers, rebuilt only when one of the pointers has moved


def _param_arrays(layers):
    """ctypes tables (one entry per layer) of everything in `layers` that belongs to the MODEL, not to the batch: built once and reused while
    the tensors keep their addresses (40 data_ptr() reads a step instead of a dozen table constructions)."""
    import ctypes as C
    names = ("We", "be", "Wj", "bj", "Wi", "gamma", "beta")
    key = tuple((l[k].data_ptr() if l[k] is not None else 0) for l in layers for k in names) + tuple(
        ((bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr() if bn.track_running_stats else 0, bn.momentum, bn.eps)
         if bn is not None else None) for bn in (l["bn"] for l in layers))
    hit = _PARAM_ARRAYS.get(len(layers))
    if hit is not None and hit[0] == key:
        return hit[1]
    t = {k: _parr([l[k] for l in layers]) for k in names}
    bns = [l["bn"] for l in layers]
    t["rm"] = _parr([(bn.running_mean if bn is not None else None) for bn in bns])
    t["rv"] = _parr([(bn.running_var if bn is not None else None) for bn in bns])
    t["nbt"] = _parr([(bn.num_batches_tracked if bn is not None and bn.track_running_stats else None) for bn in bns])
    t["momentum"] = _iarr([(bn.momentum if bn is not None else 0.0) for bn in bns], C.c_float)
    t["eps"] = _iarr([(bn.eps if bn is not None else 0.0) for bn in bns], C.c_float)
    _PARAM_ARRAYS[len(layers)] = (key, t)
    return t


@on_device_of
def static_train_fwd(x0, layers):
    """`layers`: list of dicts (one per conv layer, then optionally the decoder's Linear + BN block with plan None) with keys
    plan_parts (rowptr, src, eid) | None, n_dst, edge_attr | None, We, be, Wj, bj, Wi, gamma, beta, bn.
    -> (y_last, buf, meta): one fp32 buffer holding every layer's a / z / y / stats, `meta` the element offsets."""
    import ctypes as C
    _req(x0, "x", dim=2)
    dev, L = x0.device, len(layers)
    widths = [x0.size(1)] + [l["Wj"].size(0) for l in layers]
    meta, off = [], 0
    for i, l in enumerate(layers):
        n, ci, co = l["n_dst"], widths[i], widths[i + 1]
        if l["Wj"].size(1) != ci:
            raise ValueError("layer %d: lin_j %s does not take %d channels" % (i, tuple(l["Wj"].shape), ci))
        m = dict(a=off if l["plan_parts"] is not None else None)
        off += n * ci if l["plan_parts"] is not None else 0
        if l["bn"] is None:      # a plain Linear (the decoder's output layer): y only
            m["z"] = m["stats"] = None
            m["y"], off = off, off + n * co
        else:
            m["z"], off = off, off + n * co
            m["y"], off = off, off + n * co
            m["stats"], off = off, off + 4 * co
        meta.append(m)
    buf = _f32_work(off, dev)
    base = buf.data_ptr()
    at = lambda o: None if o is None else base + 4 * o
    scratch = _f32_work(max(lib().dgnn_colstats_scratch_elems(l["n_dst"], widths[i + 1]) for i, l in enumerate(layers)) + 2, dev)
    pp = lambda k: _parr([(l["plan_parts"][k] if l["plan_parts"] is not None else None) for l in layers])
    key = lambda k: _parr([l[k] for l in layers])
    f_e = max([l["We"].size(1) for l in layers if l["We"] is not None] or [0])
    pa = _param_arrays(layers)
    check(lib().dgnn_static_train_fwd(
        L, pp(0), pp(1), pp(2), _iarr([l["n_dst"] for l in layers], C.c_int64), ptr(x0), _ld(x0), _iarr(widths, C.c_int32),
        key("edge_attr"), _iarr([(_ld(l["edge_attr"]) if l["edge_attr"] is not None else 0) for l in layers], C.c_int64), f_e,
        pa["We"], pa["be"], pa["Wj"], pa["bj"], pa["Wi"], pa["gamma"], pa["beta"], pa["rm"], pa["rv"], pa["nbt"], pa["momentum"], pa["eps"],
        _parr([at(m["a"]) for m in meta]), _parr([at(m["z"]) for m in meta]), _parr([at(m["stats"]) for m in meta]), _parr([at(m["y"]) for m in meta]),
        ptr(scratch), GEMM_MODE, stream_ptr()), "dgnn_static_train_fwd")
    _written_behind_torch(*[t for l in layers if l["bn"] is not None
                            for t in (l["bn"].running_mean, l["bn"].running_var, l["bn"].num_batches_tracked if l["bn"].track_running_stats else None)])
    n, co = layers[-1]["n_dst"], widths[-1]
    return torch.as_strided(buf, (n, co), (co, 1), meta[-1]["y"]), buf, (meta, widths, pa)


@on_device_of
def static_train_bwd(x0, layers, buf, meta_widths, dy, keep=None):
    """-> per-layer parameter gradients [(dWe, dbe, dWj, dbj, dWi, dgamma, dbeta), ...] (views of one buffer; None where the layer has
    no such parameter).  `keep`: a dict owned by the caller (the model) -- the gradient buffer and its views are then made ONCE and every step writes
    into the same tensors (round 6: one allocation and 34 view ops a step less; what the optimizer reads through p.grad stays at one address)."""
    import ctypes as C
    meta, widths, pa = meta_widths
    dev, L = x0.device, len(layers)
    base = buf.data_ptr()
    at = lambda o: None if o is None else base + 4 * o
    kept = keep.get("static_bwd") if keep is not None else None
    sig = (dev, tuple(widths), tuple((l["We"].size(1) if l["We"] is not None else 0, l["bj"] is not None, l["Wi"] is not None, l["bn"] is not None) for l in layers))
    if kept is not None and kept[0] == sig:
        _, sizes, flat, grads_kept = kept
    else:
        sizes, off = [], 0
        for i, l in enumerate(layers):
            ci, co = widths[i], widths[i + 1]
            fe = l["We"].size(1) if l["We"] is not None else 0
            row = []
            nbn = co if l["bn"] is not None else 0
            for sz in (ci * fe, ci if fe else 0, co * ci, co if l["bj"] is not None else 0, co * ci if l["Wi"] is not None else 0, nbn, nbn):
                row.append((off, sz) if sz else None)
                off += sz
            sizes.append(row)
        flat = torch.empty(off, dtype=torch.float32, device=dev)
        grads_kept = None
    gbase = flat.data_ptr()
    gat = lambda e: None if e is None else gbase + 4 * e[0]
    n_src = [(l["n_src"] if l["plan_parts"] is not None else l["n_dst"]) for l in layers]
    dxn = max([n_src[i] * widths[i] for i in range(1, L)] or [1])
    dxn = (dxn + (1 << 20) - 1) >> 20 << 20
    dxb = torch.empty((2, dxn), dtype=torch.float32, device=dev)
    f_e = max([l["We"].size(1) for l in layers if l["We"] is not None] or [0])
    scratch = _f32_work(lib().dgnn_static_train_scratch_elems(L, _iarr(n_src, C.c_int64), _iarr([l["n_dst"] for l in layers], C.c_int64),
                                                              _iarr(widths, C.c_int32), f_e), dev)
    tp = lambda k: _parr([(l["t_parts"][k] if l["plan_parts"] is not None else None) for l in layers])
    key = lambda k: _parr([l[k] for l in layers])
    col = lambda j: _parr([gat(r[j]) for r in sizes])
    check(lib().dgnn_static_train_bwd(
        L, tp(0), tp(1), tp(2), _parr([(l["plan_parts"][0] if l["plan_parts"] is not None else None) for l in layers]), _iarr(n_src, C.c_int64),
        _iarr([l["n_dst"] for l in layers], C.c_int64), ptr(x0), _ld(x0), _iarr(widths, C.c_int32), key("edge_attr"),
        _iarr([(_ld(l["edge_attr"]) if l["edge_attr"] is not None else 0) for l in layers], C.c_int64), f_e, pa["We"], pa["be"], pa["Wj"], pa["Wi"],
        pa["gamma"], _parr([at(m["stats"]) for m in meta]), pa["eps"], _parr([at(m["a"]) for m in meta]),
        _parr([at(m["z"]) for m in meta]), _parr([at(m["y"]) for m in meta]), ptr(dy), col(0), col(1), col(2), col(3), col(4), col(5), col(6),
        _parr([dxb[0], dxb[1]]), ptr(scratch), GEMM_MODE, stream_ptr()), "dgnn_static_train_bwd")
    if grads_kept is not None:
        return grads_kept
    grads = []
    st = torch.as_strided      # one view op per gradient (a slice + .view pair costs 2.5 x as much on the host, 34 times a step)
    for i, (l, row) in enumerate(zip(layers, sizes)):
        ci, co = widths[i], widths[i + 1]
        fe = l["We"].size(1) if l["We"] is not None else 0
        m = lambda e, r, c: None if e is None else st(flat, (r, c), (c, 1), e[0])
        v = lambda e, n: None if e is None else st(flat, (n,), (1,), e[0])
        grads.append((m(row[0], ci, fe), v(row[1], ci), m(row[2], co, ci), v(row[3], co), m(row[4], co, ci), v(row[5], co), v(row[6], co)))
    if keep is not None:
        keep["static_bwd"] = (sig, sizes, flat, grads)
    return grads
