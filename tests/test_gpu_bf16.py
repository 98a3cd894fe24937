"""bf16 STORAGE path (BASELINE config 3): activations / edge embeddings bf16 in HBM, products on the bf16
matrix cores, fp32 accumulation, fp32 parameters.

Round 4: in the default (compensated) arithmetic the last layer's launch carries the decoder (its rows are never stored) and the rows of the other
layers are stored UNSIGNED (post-ReLU rows have no sign: 9 significant bits in the same 16, ops.UROWS).  On the 1M-tet metric graph SURVEY 8c's
flat bound then holds for EVERY logit: max |dlogit| 2.1e-2 <= 5e-2 (rms 2.1e-3, arg-max agreement 99.97 %) -- asserted by
test_inference_layer_bf16_metric_graph_and_ignatius; the whole Ignatius scene (logits up to +-167: 5e-2 absolute would be 3e-4 relative, below what
any 16-bit row can carry) is held to the relative form below.  With DGNN_BF16_UNSIGNED_ROWS=0 / DGNN_FUSE_DECODER=0 (the round-3 path) and in the
single-product mode the two-level statement applies:

Stated tolerance on logits against the fp32 reference (SURVEY 8c asks for 5e-2 abs and >= 99.9 % arg-max agreement):
  * |dlogit| <= 5e-2 * max(1, |logit| / 8) for at least 99.99 % of the logits, and <= 1e-1 * max(1, |logit| / 8) for every one
    -- measured on the 1M-tet metric graph (2 M logits) in the default compensated mode: rms 5.7e-3, 99.9th percentile 2.6e-2,
    maximum 5.2e-2.  The maximum over millions of cells is a tail event of the STORAGE rounding itself (2^-9 per stored value
    per layer, which no arithmetic inside the kernels can remove), hence the two-level statement;
  * arg-max (the in/out label) agrees on >= 99.9 % of the cells and on every cell whose margin exceeds the tolerance.
The plain single-product mode (DGNN_BF16_MODE=single) is ~2x further out (rms 1.0e-2, max 1.1e-1, 99.86 % agreement) and is
held to 2x these bounds.  Kernel-level checks compare each bf16 op with the SAME op evaluated in fp64 on the inputs as the
kernel sees them, so those bounds are tight."""
import os
import numpy as np
import pytest
import torch

from dgnn_amd.config import Config
from helpers import f3_data, gold, oracle_static
from test_gpu_parity import DEV, hip_static, rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
EPS = 2.0 ** -8   # bf16 spacing relative to the value (8 significant bits): rounding error <= EPS/2


def bf_check(got, ref, what="", min_agree=0.999):
    """the stated bf16 tolerance on logits (module docstring).  On a scene with fewer than 10^4 cells a single near-tie flip is
    more than 0.1 %: there the label must agree wherever the margin exceeds the tolerance and `min_agree` is lowered by the
    caller."""
    from dgnn_amd import ops
    k = 1.0 if ops.BF16_MODE == ops.BF16_COMPENSATED else 2.0
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    tol = k * 5e-2 * np.maximum(1.0, np.abs(ref) / 8)
    err = np.abs(got - ref)
    agree = float((got.argmax(1) == ref.argmax(1)).mean())
    margin = np.abs(ref[:, 0] - ref[:, 1]) > 2 * tol.max(axis=1)
    flips = int((got.argmax(1)[margin] != ref.argmax(1)[margin]).sum())
    inside = float((err <= tol).mean())
    print("%s bf16: max|dlogit| %.3e, rms %.3e, p99.9 %.3e, within tol %.6f, arg-max agreement %.5f (%d flips above margin)" % (
        what, err.max(), np.sqrt((err ** 2).mean()), np.percentile(err, 99.9), inside, agree, flips))
    assert inside >= 0.9999 and (err <= 2 * tol).all(), "%s max|dlogit| %.3e, within tolerance %.6f" % (what, err.max(), inside)
    assert flips == 0 and agree >= (min_agree if k == 1.0 else min(min_agree, 0.998)), "%s arg-max agreement %.5f, %d flips above the margin" % (what, agree, flips)
    return float(err.max()), agree


def test_cast_round_trip_and_padding():
    from dgnn_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1001, 29, generator=g) * 50
    xb = ops.cast_to_bf16(x.to(DEV)[:, 1:])                 # strided fp32 view in, packed bf16 out
    assert xb.dtype == BF and xb.shape == (1001, 28)
    assert torch.equal(xb.cpu(), x[:, 1:].to(BF))           # round to nearest even, as torch
    xp = ops.cast_to_bf16(x.to(DEV), 32)
    assert xp.shape == (1001, 32) and torch.equal(xp[:, :29].cpu(), x.to(BF)) and bool((xp[:, 29:] == 0).all())
    assert torch.equal(ops.cast_to_f32(xb).cpu(), x[:, 1:].to(BF).float())


@pytest.mark.parametrize("mode", ["compensated", "single"])
@pytest.mark.parametrize("c_in,c_out", [(28, 64), (64, 128), (128, 128), (64, 64), (32, 128)])
def test_fused_layer_bf16_vs_fp64_on_rounded_inputs(c_in, c_out, mode):
    """One fused bf16 layer against fp64.  "single": every operand rounded to bf16 once -> reference on the bf16-rounded x,
    attributes, parameters and mean.  "compensated" (default): only the stored rows are bf16 -> reference on the rounded x and
    the exact fp32 attributes / parameters / mean; what remains is the 2^-16 of the (hi, lo) pairs and the rounding of the
    stored result."""
    from dgnn_amd import ops
    from dgnn_amd.synthetic import delaunay_tet_graph
    from oracle.pyg_semantics import propagate_mean
    adj, _, _ = delaunay_tet_graph(900, seed=c_in)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64))
    g = torch.Generator().manual_seed(c_in + c_out)
    x = torch.randn(n, c_in, generator=g)
    x[::53] *= 30
    ea = torch.randn(4 * n, 20, generator=g)
    We, be = torch.randn(c_in, 20, generator=g) * 0.3, torch.randn(c_in, generator=g)
    Wj, Wi, bj = torch.randn(c_out, c_in, generator=g) * 0.1, torch.randn(c_out, c_in, generator=g) * 0.1, torch.randn(c_out, generator=g)
    scale, shift = torch.rand(c_out, generator=g) + 0.5, torch.randn(c_out, generator=g)
    r = lambda t: t.to(BF).double()
    comp = mode == "compensated"
    p = (lambda t: t.double()) if comp else r                   # parameters / attributes as the matrix cores see them
    phi = p(ea) @ p(We).t() + p(be)
    a = propagate_mean(r(x), n, ei, phi)
    ab = a if comp else a.float().to(BF).double()                # "single" rounds the mean to bf16 before the dense part
    pre = ab @ p(Wj).t() + r(x) @ p(Wi).t() + bj.double()
    ref = torch.relu(pre * scale.double() + shift.double())
    mag = (ab.abs() @ p(Wj).abs().t() + r(x).abs() @ p(Wi).abs().t()) * scale.double().abs()   # sum of |terms|
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n, 1, n_other=n)
    xb = ops.cast_to_bf16(x.to(DEV))
    old, ops.BF16_MODE = ops.BF16_MODE, ops.BF16_COMPENSATED if comp else ops.BF16_SINGLE
    try:
        out = ops.sage_layer_fused_fwd_bf16(rowptr, src, n, xb, c_in, ea.to(DEV), We.to(DEV), be.to(DEV), Wj.to(DEV), bj.to(DEV), Wi.to(DEV),
                                            scale.to(DEV), shift.to(DEV), True, eid=eid)
        # rows in plan order (eid=None) give the same bits
        out2 = ops.sage_layer_fused_fwd_bf16(rowptr, src, n, xb, c_in, ops.gather_rows(ea.to(DEV), eid), We.to(DEV), be.to(DEV), Wj.to(DEV),
                                             bj.to(DEV), Wi.to(DEV), scale.to(DEV), shift.to(DEV), True)
    finally:
        ops.BF16_MODE = old
    assert out.dtype == BF and out.shape == (n, c_out)
    err = (out.cpu().double() - ref).abs()
    # bf16 spacing is 2^-8 .. 2^-7 of the value: the stored result is off by up to EPS |ref|.  single: the fp32 mean may round
    # to the other bf16 neighbour than the fp64 mean (2 EPS |a| through Wj).  compensated: (hi, lo) pairs carry 16 bits and the
    # lo x lo terms are dropped: 2^-15 of the sum of |terms|, plus fp32 accumulation.
    bound = EPS * ref.abs() + (2.0 ** -14 * mag if comp else 2 * EPS * (ab.abs() @ p(Wj).abs().t()) * scale.double().abs()) + 1e-3
    assert bool((err <= bound).all()), float((err / bound).max())
    assert torch.equal(out, out2)


@pytest.mark.parametrize("mode", ["compensated", "single"])
@pytest.mark.parametrize("c_out", [64, 128])
def test_first_layer_reads_fp32_rows_in_place(c_out, mode):
    """c_in <= 32: the fused bf16 layer takes the caller's fp32 feature rows (the reference's x[:, 1:] view: row stride 29, 4-byte
    aligned) -- no bf16 copy of the input, outliers of 170 sigma are not rounded.  Output bf16."""
    from dgnn_amd import ops
    from dgnn_amd.synthetic import delaunay_tet_graph
    from oracle.pyg_semantics import propagate_mean
    adj, _, _ = delaunay_tet_graph(700, seed=c_out)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64))
    g = torch.Generator().manual_seed(c_out)
    x29 = torch.randn(n, 29, generator=g)
    x29[::41] *= 170
    x = x29[:, 1:]
    ea = torch.randn(4 * n, 20, generator=g)
    We, be = torch.randn(28, 20, generator=g) * 0.3, torch.randn(28, generator=g)
    Wj, Wi, bj = torch.randn(c_out, 28, generator=g) * 0.1, torch.randn(c_out, 28, generator=g) * 0.1, torch.randn(c_out, generator=g)
    r = lambda t: t.to(BF).double()
    comp = mode == "compensated"
    p = (lambda t: t.double()) if comp else r
    a = propagate_mean(x.double(), n, ei, p(ea) @ p(We).t() + p(be))        # gathered rows are exact fp32 in both modes
    ab = a if comp else a.float().to(BF).double()
    xi = x.double() if comp else r(x)                                       # own row: (hi, lo) pair / rounded once
    ref = ab @ p(Wj).t() + xi @ p(Wi).t() + bj.double()
    mag = ab.abs() @ p(Wj).abs().t() + xi.abs() @ p(Wi).abs().t()
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n, 1, n_other=n)
    xd = x29.to(DEV)[:, 1:]
    assert ops.fused_layer_supported_bf16(28, c_out, 20, xd)
    old, ops.BF16_MODE = ops.BF16_MODE, ops.BF16_COMPENSATED if comp else ops.BF16_SINGLE
    try:
        out = ops.sage_layer_fused_fwd_bf16(rowptr, src, n, xd, 28, ea.to(DEV), We.to(DEV), be.to(DEV), Wj.to(DEV), bj.to(DEV), Wi.to(DEV),
                                            None, None, False, eid=eid)
    finally:
        ops.BF16_MODE = old
    assert out.dtype == BF
    err = (out.cpu().double() - ref).abs()
    bound = EPS * ref.abs() + (2.0 ** -14 * mag if comp else 2 * EPS * (ab.abs() @ p(Wj).abs().t())) + 1e-3
    assert bool((err <= bound).all()), float((err / bound).max())


def test_fused_layer_bf16_irregular_degrees_and_tail():
    """in-degree 0..many (generic per-lane path) and a destination count that is not a multiple of the tile"""
    from dgnn_amd import ops
    from oracle.pyg_semantics import propagate_mean
    g = torch.Generator().manual_seed(5)
    n_src, n_dst, E, c_in, c_out = 1500, 1003, 5200, 64, 128
    ei = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, n_dst + 30, (E,), generator=g).clamp_max(n_dst - 1)])
    x = torch.randn(n_src, c_in, generator=g)
    ea = torch.randn(E, 20, generator=g)
    We, be = torch.randn(c_in, 20, generator=g) * 0.3, torch.randn(c_in, generator=g)
    Wj, Wi, bj = torch.randn(c_out, c_in, generator=g) * 0.1, torch.randn(c_out, c_in, generator=g) * 0.1, torch.randn(c_out, generator=g)
    r = lambda t: t.to(BF).double()
    a = propagate_mean(r(x), n_dst, ei, r(ea) @ r(We).t() + r(be))
    ref = a.float().to(BF).double() @ r(Wj).t() + r(x)[:n_dst] @ r(Wi).t() + bj.double()
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n_dst, 1, n_other=n_src)
    out = ops.sage_layer_fused_fwd_bf16(rowptr, src, n_dst, ops.cast_to_bf16(x.to(DEV)), c_in, ea.to(DEV), We.to(DEV), be.to(DEV), Wj.to(DEV),
                                        bj.to(DEV), Wi.to(DEV), None, None, False, eid=eid)
    assert rel_err(out.float(), ref) < 1.5e-2


@pytest.mark.parametrize("points,c_in", [(5, 128), (40, 128), (700, 128), (700, 96), (5000, 128)])
def test_last_bf16_layer_with_the_decoder_inside(points, c_in):
    """dgnn_sage_layer_fused_decoder_fwd_bf16 (round 4): the last conv layer of the bf16 storage path with Linear-BN-ReLU-Linear inside its launch
    (reference :180-187 applied at :350-351).  The layer's output is never rounded to bf16, so the logits must follow the fp64 evaluation of
    layer + decoder on the bf16 INPUT rows as the kernel sees them to the 2^-15 of the (hi, lo) operand pairs -- several times closer than the
    two-launch form, which stores the rows as bf16 in between; graphs smaller than a tile / not a multiple of one; destination sub-ranges bit-identical."""
    from dgnn_amd import ops
    from dgnn_amd.synthetic import delaunay_tet_graph
    from oracle.pyg_semantics import propagate_mean
    if ops.BF16_MODE != ops.BF16_COMPENSATED or not ops.FUSE_DECODER:
        pytest.skip("the one-launch form exists for the compensated arithmetic (DGNN_FUSE_DECODER=0 selects the two-launch form)")
    adj, _, _ = delaunay_tet_graph(points, seed=points)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64))
    g = torch.Generator().manual_seed(points + c_in)
    x = torch.relu(torch.randn(n, c_in, generator=g))           # post-ReLU activations, as the layer sees them
    x[::53] *= 20
    ea = torch.randn(4 * n, 20, generator=g)
    We, be = torch.randn(c_in, 20, generator=g) * 0.3, torch.randn(c_in, generator=g)
    Wj, Wi, bj = torch.randn(128, c_in, generator=g) * 0.1, torch.randn(128, c_in, generator=g) * 0.1, torch.randn(128, generator=g)
    scale, shift = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    W0, b0 = torch.randn(64, 128, generator=g) * 0.15, torch.randn(64, generator=g)
    s1, h1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    W3, b3 = torch.randn(2, 64, generator=g) * 0.3, torch.randn(2, generator=g)
    r = lambda t: t.to(BF).double()
    d = lambda t: t.double()
    a = propagate_mean(r(x), n, ei, d(ea) @ d(We).t() + d(be))
    y = torch.relu((a @ d(Wj).t() + r(x) @ d(Wi).t() + d(bj)) * d(scale) + d(shift))
    hid = torch.relu((y @ d(W0).t() + d(b0)) * d(s1) + d(h1))
    ref = hid @ d(W3).t() + d(b3)
    mag = (hid.abs() @ d(W3).abs().t()) + (y.abs() @ d(W0).abs().t() * d(s1).abs()) @ d(W3).abs().t()     # sum of |terms| behind a logit
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n, 1, n_other=n)
    xb = ops.cast_to_bf16(x.to(DEV))
    dv = lambda *ts: [t.to(DEV) for t in ts]
    layer = dv(We, be, Wj, bj, Wi, scale, shift)
    deco = dv(W0, b0, s1, h1, W3, b3)
    assert ops.fused_layer_decoder_supported_bf16(c_in, 128, 20, 64, 2, xb)
    one = ops.sage_layer_fused_decoder_fwd_bf16(rowptr, src, n, xb, c_in, ea.to(DEV), *layer, True, *deco, eid=eid)
    assert one.dtype == torch.float32 and one.shape == (n, 2)
    two = ops.decoder_fused_fwd_bf16(ops.sage_layer_fused_fwd_bf16(rowptr, src, n, xb, c_in, ea.to(DEV), *layer, True, eid=eid), *deco)
    e1, e2 = (one.cpu().double() - ref).abs(), (two.cpu().double() - ref).abs()
    bound = 2.0 ** -13 * mag + 1e-4
    assert bool((e1 <= bound).all()), float((e1 / bound).max())
    if n >= 1000:
        assert e1.pow(2).mean().sqrt() < 0.2 * e2.pow(2).mean().sqrt(), (float(e1.max()), float(e2.max()))     # the bf16 round trip of the rows is what the two launches pay
    assert torch.equal(one, ops.sage_layer_fused_decoder_fwd_bf16(rowptr, src, n, xb, c_in, ea.to(DEV), *layer, True, *deco, eid=eid))     # run to run
    # rows already in plan order (eid = None), and destination sub-ranges (interior / boundary launches of a partitioned scene): the same bits
    assert torch.equal(one, ops.sage_layer_fused_decoder_fwd_bf16(rowptr, src, n, xb, c_in, ops.gather_rows(ea.to(DEV), eid), *layer, True, *deco))
    if n > 40:
        out = torch.full((n, 2), float("nan"), device=DEV)
        for b, e in ((0, 37), (37, n - 5), (n - 5, n)):
            ops.sage_layer_fused_decoder_fwd_bf16(rowptr[b:e + 1], src, e - b, xb, c_in, ea.to(DEV), *layer, True, *deco, out=out[b:e], eid=eid, x_dst=xb[b:e])
        assert torch.equal(out, one)


def test_last_bf16_layer_with_the_decoder_inside_irregular_degrees():
    """in-degrees 0 .. many (the per-lane generic path inside the launch) and a ragged destination count"""
    from dgnn_amd import ops
    g = torch.Generator().manual_seed(11)
    n_src, n_dst, E, c_in = 1500, 1003, 5200, 128
    ei = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, n_dst + 30, (E,), generator=g).clamp_max(n_dst - 1)])
    x = torch.relu(torch.randn(n_src, c_in, generator=g))
    ea = torch.randn(E, 20, generator=g)
    We, be = torch.randn(c_in, 20, generator=g) * 0.3, torch.randn(c_in, generator=g)
    Wj, Wi, bj = torch.randn(128, c_in, generator=g) * 0.1, torch.randn(128, c_in, generator=g) * 0.1, torch.randn(128, generator=g)
    W0, b0 = torch.randn(64, 128, generator=g) * 0.15, torch.randn(64, generator=g)
    W3, b3 = torch.randn(2, 64, generator=g) * 0.3, torch.randn(2, generator=g)
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n_dst, 1, n_other=n_src)
    xb = ops.cast_to_bf16(x.to(DEV))
    dv = lambda *ts: [t.to(DEV) if t is not None else None for t in ts]
    layer, deco = dv(We, be, Wj, bj, Wi, None, None), dv(W0, b0, None, None, W3, b3)
    one = ops.sage_layer_fused_decoder_fwd_bf16(rowptr, src, n_dst, xb, c_in, ea.to(DEV), *layer, True, *deco, eid=eid)
    rows = ops.sage_layer_fused_fwd_bf16(rowptr, src, n_dst, xb, c_in, ea.to(DEV), *layer, True, eid=eid)
    two = ops.decoder_fused_fwd_bf16(rows, *deco)
    # the two-launch form sees the rows rounded to bf16: one relative 2^-9 per row element through 128 -> 64 -> 2
    y = rows.float().cpu().double()
    mag = ((y.abs() @ W0.double().abs().t()) @ W3.double().abs().t())
    assert bool(((one.cpu().double() - two.cpu().double()).abs() <= EPS * mag + 1e-3).all())


def _ub_decode(t):
    """unsigned rows (int16 container) -> float64 values: bits << 15"""
    b = (t.cpu().to(torch.int32) & 0xFFFF) << 15
    return b.view(torch.float32).double()


def _ub_round(t):
    """fp32 >= 0 -> the value its unsigned 16-bit row keeps (round to nearest even on bit 15)"""
    b = t.contiguous().view(torch.int32)
    b = ((b + 0x3FFF + ((b >> 15) & 1)) >> 15) << 15
    return b.view(torch.float32)


@pytest.mark.parametrize("c_in,c_out", [(28, 64), (64, 128), (128, 128), (64, 64), (96, 128)])
def test_unsigned_rows_fused_layer_vs_fp64(c_in, c_out):
    """UNSIGNED 16-bit rows (ops.UROWS, round 4): a layer on fp32 feature rows starts the format, a layer on unsigned rows keeps it.  Against fp64 on
    the inputs as the kernel sees them: the stored result is off by at most HALF of bf16's spacing (2^-10 of the value) + the (hi, lo) operand terms."""
    from dgnn_amd import ops
    from dgnn_amd.synthetic import delaunay_tet_graph
    from oracle.pyg_semantics import propagate_mean
    if ops.BF16_MODE != ops.BF16_COMPENSATED:
        pytest.skip("unsigned rows belong to the compensated arithmetic")
    adj, _, _ = delaunay_tet_graph(900, seed=c_in)
    n = adj.shape[0] // 4
    ei = torch.from_numpy(adj.T.astype(np.int64))
    g = torch.Generator().manual_seed(c_in * 7 + c_out)
    first = c_in <= 32
    x = torch.randn(n, c_in, generator=g) if first else torch.relu(torch.randn(n, c_in, generator=g))
    x[::53] *= 30
    ea = torch.randn(4 * n, 20, generator=g)
    We, be = torch.randn(c_in, 20, generator=g) * 0.3, torch.randn(c_in, generator=g)
    Wj, Wi, bj = torch.randn(c_out, c_in, generator=g) * 0.1, torch.randn(c_out, c_in, generator=g) * 0.1, torch.randn(c_out, generator=g)
    scale, shift = torch.rand(c_out, generator=g) + 0.5, torch.randn(c_out, generator=g)
    d = lambda t: t.double()
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n, 1, n_other=n)
    if first:
        xin, xd = x.to(DEV), x.double()                                    # fp32 rows, read in place
    else:
        xr = _ub_round(x)
        xin = ((xr.view(torch.int32) >> 15).to(torch.int16)).to(DEV)       # the rows as a previous layer stores them
        xd = xr.double()
        assert torch.equal(_ub_decode(xin), xd)
        assert torch.equal(ops.rows_unsigned_to_bf16(xin).cpu(), xr.to(BF))  # converter: round to nearest even to plain bf16
    a = propagate_mean(xd, n, ei, d(ea) @ d(We).t() + d(be))
    ref = torch.relu((a @ d(Wj).t() + xd @ d(Wi).t() + d(bj)) * d(scale) + d(shift))
    mag = (a.abs() @ d(Wj).abs().t() + xd.abs() @ d(Wi).abs().t()) * d(scale).abs()
    dv = lambda *ts: [t.to(DEV) for t in ts]
    out = ops.sage_layer_fused_fwd_bf16(rowptr, src, n, xin, c_in, ea.to(DEV), *dv(We, be, Wj, bj, Wi, scale, shift), True, eid=eid, rows_out_unsigned=True)
    assert out.dtype == ops.UROWS and out.shape == (n, c_out)
    err = (_ub_decode(out) - ref).abs()
    # an unsigned OWN row enters the matrix cores rounded to bf16 (its ninth bit is not worth a sixth product per k-step: csrc/fused_bf16.hip XPARTS);
    # the gathered rows keep all nine bits
    own = torch.zeros_like(ref) if first else 0.5 * EPS * (xd.abs() @ d(Wi).abs().t()) * d(scale).abs()
    bound = 0.5 * EPS * ref.abs() + 2.0 ** -14 * mag + own + 1e-3
    assert bool((err <= bound).all()), float((err / bound).max())
    # the same layer writing plain bf16 rows (fp32 input only: a layer on 16-bit rows keeps its input's format) is twice as far out at the worst row
    if first:
        plain = ops.sage_layer_fused_fwd_bf16(rowptr, src, n, xin, c_in, ea.to(DEV), *dv(We, be, Wj, bj, Wi, scale, shift), True, eid=eid)
        assert plain.dtype == BF
        e_plain = (plain.cpu().double() - ref).abs()
        assert err.pow(2).mean().sqrt() < 0.7 * e_plain.pow(2).mean().sqrt()
    else:
        with pytest.raises(Exception):
            ops.sage_layer_fused_fwd_bf16(rowptr, src, n, xin, c_in, ea.to(DEV), *dv(We, be, Wj, bj, Wi, scale, shift), True, eid=eid, rows_out_unsigned=False)
    if c_in > 64 and c_out == 128:      # ... and into the decoder-carrying launch
        W0, b0 = torch.randn(64, 128, generator=g) * 0.15, torch.randn(64, generator=g)
        W3, b3 = torch.randn(2, 64, generator=g) * 0.3, torch.randn(2, generator=g)
        hid = torch.relu(ref @ d(W0).t() + d(b0))
        lref = hid @ d(W3).t() + d(b3)
        lmag = (hid.abs() @ d(W3).abs().t()) + (ref.abs() @ d(W0).abs().t()) @ d(W3).abs().t()
        lg = ops.sage_layer_fused_decoder_fwd_bf16(rowptr, src, n, xin, c_in, ea.to(DEV), *dv(We, be, Wj, bj, Wi, scale, shift), True,
                                                   *dv(W0, b0), None, None, *dv(W3, b3), eid=eid)
        e = (lg.cpu().double() - lref).abs()
        lb = 2.0 ** -13 * lmag + (own @ d(W0).abs().t()) @ d(W3).abs().t() + 1e-4
        assert bool((e <= lb).all()), float((e / lb).max())


def test_generic_bf16_ops_vs_fp64_on_rounded_inputs():
    """aggregate (fused filter / given phi) fwd + bwd, GEMM fwd, weight gradient, BatchNorm train fwd/bwd, relu, column sums"""
    from dgnn_amd import ops
    from oracle.pyg_semantics import propagate_mean
    g = torch.Generator().manual_seed(11)
    n_src, n_dst, E, c = 800, 600, 2900, 96
    ei = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, n_dst, (E,), generator=g)])
    x, phi, ea = torch.randn(n_src, c, generator=g), torch.randn(E, c, generator=g), torch.randn(E, 20, generator=g)
    We, be = torch.randn(c, 20, generator=g) * 0.3, torch.randn(c, generator=g)
    r = lambda t: t.to(BF).double()
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n_dst, 1, n_other=n_src)
    xb, pb = ops.cast_to_bf16(x.to(DEV)), ops.cast_to_bf16(phi.to(DEV))
    a = ops.aggregate_fwd(rowptr, src, eid, n_dst, xb, phi=pb)
    assert a.dtype == BF and rel_err(a.float(), propagate_mean(r(x), n_dst, ei, r(phi))) < EPS
    a2, phi_o = ops.aggregate_fwd(rowptr, src, eid, n_dst, xb, ea.to(DEV), We.to(DEV), be.to(DEV), want_phi=True)
    phi64 = ea.double() @ We.double().t() + be.double()
    assert rel_err(a2.float(), propagate_mean(r(x), n_dst, ei, phi64)) < EPS and rel_err(phi_o.float(), phi64) < EPS
    # backward (given phi): dx, dphi against autograd in fp64
    xd, pd = r(x).requires_grad_(), r(phi).requires_grad_()
    da = torch.randn(n_dst, c, generator=g)
    (propagate_mean(xd, n_dst, ei, pd) * r(da)).sum().backward()
    t_rowptr, t_dst, t_eid = ops.plan_build(ei.to(DEV), n_src, 0, n_other=n_dst)
    dx, _, _, dphi = ops.aggregate_bwd(t_rowptr, t_dst, t_eid, n_src, rowptr, xb, ops.cast_to_bf16(da.to(DEV)), phi=pb)
    assert dx.dtype == BF and rel_err(dx.float(), xd.grad) < EPS and rel_err(dphi.float(), pd.grad) < EPS
    dx2, dWe, dbe, _ = ops.aggregate_bwd(t_rowptr, t_dst, t_eid, n_src, rowptr, xb, ops.cast_to_bf16(da.to(DEV)), ea.to(DEV), We.to(DEV), be.to(DEV))
    Wd, bd, xd2 = We.double().requires_grad_(), be.double().requires_grad_(), r(x).requires_grad_()
    (propagate_mean(xd2, n_dst, ei, ea.double() @ Wd.t() + bd) * r(da)).sum().backward()
    assert rel_err(dx2.float(), xd2.grad) < EPS and rel_err(dWe, Wd.grad) < 2e-5 and rel_err(dbe, bd.grad) < 2e-5
    # GEMM: two bf16 operands, fp32 weights (rounded to bf16 when staged), bias, scale/shift, ReLU; bf16 and fp32 outputs
    for M, k1, k2, n_out in [(1000, 28, 28, 64), (777, 64, 64, 128), (2048, 128, 128, 128), (300, 128, 0, 2), (130, 38, 6, 70), (1, 128, 0, 64),
                             (515, 256, 256, 512), (4000, 2, 0, 28)]:
        A1, W1 = torch.randn(M, k1, generator=g), torch.randn(n_out, k1, generator=g) * 0.2
        A2, W2 = (torch.randn(M, k2, generator=g), torch.randn(n_out, k2, generator=g) * 0.2) if k2 else (None, None)
        bias, sc, sh = torch.randn(n_out, generator=g), torch.rand(n_out, generator=g) + 0.5, torch.randn(n_out, generator=g)
        ref = r(A1) @ r(W1).t() + bias.double()
        if k2:
            ref = ref + r(A2) @ r(W2).t()
        ref = torch.relu(ref * sc.double() + sh.double())
        dv = lambda t: None if t is None else t.to(DEV)
        bfv = lambda t: None if t is None else ops.cast_to_bf16(t.to(DEV))[:, :t.size(1)]
        for od in (BF, torch.float32):
            out = ops.linear_fwd(bfv(A1), dv(W1), bfv(A2), dv(W2), dv(bias), dv(sc), dv(sh), True, out_dtype=od)
            assert out.dtype == od
            assert rel_err(out.float(), ref) < (EPS if od == BF else 2e-6), (M, k1, k2, n_out, od)
    # weight gradient: dW = G^T . A with bf16 / fp32 operand mixes
    for M, na, nb in [(5000, 128, 128), (333, 64, 28), (70000, 2, 64), (64, 130, 38)]:
        G, A = torch.randn(M, na, generator=g), torch.randn(M, nb, generator=g)
        ref = r(G).t() @ r(A)
        Gb, Ab = ops.cast_to_bf16(G.to(DEV))[:, :na], ops.cast_to_bf16(A.to(DEV))[:, :nb]
        assert rel_err(ops.linear_wgrad(Gb, Ab), ref) < 2e-5
        assert rel_err(ops.linear_wgrad(G.to(DEV), Ab), ref) < 2e-5     # fp32 gradient of fp32 logits against bf16 activations
        assert rel_err(ops.colsum(Gb), r(G).sum(0)) < 2e-5
    # BatchNorm train-mode forward / backward on bf16 rows (statistics in fp64 on the device)
    M, c2 = 3000, 70
    xx, gg = torch.randn(M, c2, generator=g) * 3 + 1, torch.randn(M, c2, generator=g)
    gamma, beta = torch.rand(c2, generator=g) + 0.5, torch.randn(c2, generator=g)
    xr, gr = r(xx), r(gg)
    bn = torch.nn.BatchNorm1d(c2).double().train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    xin = xr.clone().requires_grad_()
    y = torch.relu(bn(xin))
    xbn = ops.cast_to_bf16(xx.to(DEV))
    mean, var = ops.bn_batch_stats(xbn)
    assert rel_err(mean, xr.mean(0)) < 1e-6 and rel_err(var, xr.var(0, unbiased=False)) < 1e-5
    s_, t_ = ops.bn_fold(gamma.to(DEV), beta.to(DEV), mean, var, 1e-5)
    yb = ops.scale_shift_act(xbn, s_, t_, True)
    assert yb.dtype == BF and rel_err(yb.float(), y) < EPS
    # backward with the bf16-rounded y as the ReLU mask source (what the kernel sees)
    (y * gr).sum().backward()
    dxb, dgam, dbet = ops.bn_relu_bwd(xbn, yb, ops.cast_to_bf16(gg.to(DEV)), gamma.to(DEV), mean, var, 1e-5, True, True)
    assert dxb.dtype == BF and rel_err(dxb.float(), xin.grad) < 2 * EPS
    assert rel_err(dgam, bn.weight.grad) < 1e-3 and rel_err(dbet, bn.bias.grad) < 1e-3
    # relu / relu_bwd
    assert torch.equal(ops.relu(xbn).cpu(), torch.relu(xx.to(BF)))
    assert torch.equal(ops.relu_bwd(yb, ops.cast_to_bf16(gg.to(DEV))).cpu(), torch.where(yb.cpu() > 0, gg.to(BF), torch.zeros((), dtype=BF)))


def _bf16_inference(net, data):
    net.set_storage_dtype(BF)
    try:
        return net.inference_layer(data).cpu().numpy()
    finally:
        net.set_storage_dtype(torch.float32)


def test_inference_layer_bf16_golden_f2_and_lognormal_inputs():
    g = gold("static_f2_regular256.npz")
    net = hip_static()
    data = Config(x=torch.from_numpy(g["x"]).to(DEV), edge_attr=torch.from_numpy(g["edge_attr"]).to(DEV),
                  edge_index=torch.from_numpy(g["adjacencies"].T.astype(np.int64)).to(DEV))
    logits = _bf16_inference(net, data)
    assert logits.dtype == np.float32
    err, agree = bf_check(logits, g["logits"], "F2", min_agree=0.99)   # 256 cells
    print("bf16 F2: max|dlogit| %.3e, arg-max agreement %.4f" % (err, agree))
    # heavy-tailed (log-normal) inputs: bf16 keeps the fp32 exponent range, nothing may overflow
    from dgnn_amd.synthetic import delaunay_tet_graph
    adj, _, _ = delaunay_tet_graph(3000, seed=8)
    n = adj.shape[0] // 4
    gen = torch.Generator().manual_seed(2)
    x = torch.exp(2.5 * torch.randn(n, 29, generator=gen)) * torch.sign(torch.randn(n, 29, generator=gen))
    ea = torch.exp(1.5 * torch.randn(4 * n, 20, generator=gen)) * torch.sign(torch.randn(4 * n, 20, generator=gen))
    ei = torch.from_numpy(adj.T.astype(np.int64))
    with torch.no_grad():
        ref = oracle_static().inference_layer(Config(x=x, edge_attr=ea, edge_index=ei)).numpy()
    got = _bf16_inference(net, Config(x=x.to(DEV), edge_attr=ea.to(DEV), edge_index=ei.to(DEV)))
    assert np.isfinite(got).all()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-2 * scale, (np.abs(got - ref).max(), scale)
    assert (got.argmax(1) == ref.argmax(1)).mean() >= 0.995


def test_inference_layer_bf16_metric_graph_and_ignatius():
    """the 1M-tet metric graph and the whole real scene in bf16 storage against the fp32 references"""
    from test_gpu_scale import metric_graph
    s = metric_graph()
    net = hip_static()
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = oracle_static().inference_layer(Config(x=s["x"], edge_attr=s["ea"], edge_index=s["ei"])).numpy()
    got = _bf16_inference(net, Config(x=s["x"].to(DEV), edge_attr=s["ea"].to(DEV), edge_index=s["ei"].to(DEV)))
    err, agree = bf_check(got, ref, "1M")
    print("bf16 1M graph: max|dlogit| %.3e, arg-max agreement %.5f" % (err, agree))
    from dgnn_amd import ops
    if ops.BF16_MODE == ops.BF16_COMPENSATED and ops.BF16_UNSIGNED_ROWS and ops.FUSE_DECODER:
        # SURVEY 8c as written: <= 5e-2 absolute for EVERY one of the 2 M logits, >= 99.9 % arg-max agreement (measured 2.1e-2 / 99.97 %)
        assert err <= 5e-2 and agree >= 0.999, (err, agree)
        net.set_storage_dtype(BF)
        try:
            assert net.activation_dtype(0) == ops.UROWS and net.activation_dtype(2) == ops.UROWS and net.fuses_decoder(3)
        finally:
            net.set_storage_dtype(torch.float32)
    g = gold("static_f4_ignatius_full.npz")
    n = g["x"].shape[0]
    fg = np.random.default_rng(int(g["fgeom_seed"])).standard_normal((4 * n, 4)).astype(np.float32)
    ea = np.concatenate([fg, g["edge_attr16"]], axis=1)
    adj = np.stack([np.repeat(np.arange(n, dtype=np.int64), 4), g["adj_dst"].astype(np.int64)])
    got = _bf16_inference(net, Config(x=torch.from_numpy(g["x"]).to(DEV), edge_attr=torch.from_numpy(ea).to(DEV),
                                      edge_index=torch.from_numpy(adj).to(DEV)))
    err, agree = bf_check(got, g["logits"], "Ignatius")
    print("bf16 Ignatius: max|dlogit| %.3e (logit range %.0f), arg-max agreement %.5f" % (err, np.abs(g["logits"]).max(), agree))


def test_bf16_other_widths_take_the_generic_pair():
    """widths the fused bf16 kernel does not cover run the bf16 aggregate + bf16-MFMA GEMM pair"""
    from dgnn_amd.synthetic import delaunay_tet_graph
    adj, _, _ = delaunay_tet_graph(1200, seed=4)
    n = adj.shape[0] // 4
    gen = torch.Generator().manual_seed(6)
    x, ea = torch.randn(n, 29, generator=gen), torch.randn(4 * n, 20, generator=gen)
    ei = torch.from_numpy(adj.T.astype(np.int64))
    convs = (64, 128, 256, 512)
    onet = oracle_static(convs=convs, load=False, seed=4)
    net = hip_static(convs=convs, sd=onet.state_dict())
    with torch.no_grad():
        ref = onet.inference_layer(Config(x=x, edge_attr=ea, edge_index=ei)).numpy()
    got = _bf16_inference(net, Config(x=x.to(DEV), edge_attr=ea.to(DEV), edge_index=ei.to(DEV)))
    assert np.abs(got - ref).max() <= 3e-2 * max(1.0, np.abs(ref).max())


def test_train_forward_backward_bf16_static_f3():
    """Static SurfaceNet.forward in train mode (BN batch statistics) + backward in bf16 storage against the fp32 golden"""
    g = gold("static_f3_train_blocks.npz")
    net = hip_static(train=True).set_storage_dtype(BF)
    d = f3_data(g)
    data = Config(all=Config(x=d.all.x.to(DEV), edge_attr=d.all.edge_attr.to(DEV)), batch_n_id=d.batch_n_id.to(DEV),
                  batch_adjs=[(a.to(DEV), e.to(DEV), s) for a, e, s in d.batch_adjs])
    logits = net(data)
    assert logits.dtype == torch.float32
    ref = g["logits"]
    assert np.abs(logits.detach().cpu().numpy() - ref).max() <= 5e-2 * max(1.0, np.abs(ref).max() / 8)
    (logits * torch.from_numpy(g["G"]).to(DEV)).sum().backward()
    # gradients: direction and size against the fp32 reference, per tensor
    for k, p in net.named_parameters():
        r_ = torch.from_numpy(g["grad." + k]).double().flatten()
        q = p.grad.double().cpu().flatten()
        assert p.grad.dtype == torch.float32
        if r_.norm() < 1e-4 * max(np.abs(g[kk]).max() for kk in g.files if kk.startswith("grad.")):
            continue   # biases ahead of a train-mode BatchNorm: analytically zero gradient
        cos = float((q @ r_) / (q.norm() * r_.norm()))
        assert cos > 0.995 and 0.9 < float(q.norm() / r_.norm()) < 1.1, (k, cos, float(q.norm() / r_.norm()))


def test_updated_variant_bf16_forward_backward():
    """surfaceNetUpdatedEdgeFilters in bf16 storage (the BASELINE config-3 model): phi written and chained in bf16"""
    from dgnn_amd.learning.surfaceNetUpdatedEdgeFilters import SurfaceNet
    g = gold("static_f3_train_blocks.npz")
    u = gold("updated_f3_blocks.npz")
    d = f3_data(g)
    for tag, name in (("plus", "sage+"), ("plain", "sage")):
        clf = Config.wrap(dict(training=dict(model_params=[int(v) for v in u[tag + ".model_params"]], model_name=name, loss="kl"),
                               features=dict(normalization_feature=1, keep_normalization_feature=0), temp=dict(device=DEV)))
        net = SurfaceNet(28, clf)
        net.load_state_dict({k[len(tag) + 7:]: torch.from_numpy(u[k]) for k in u.files if k.startswith(tag + ".param.")})
        net = net.to(DEV).set_storage_dtype(BF)
        data = Config(x=d.all.x.to(DEV), edge_attr=d.all.edge_attr.to(DEV), n_id=d.batch_n_id.to(DEV),
                      adjs=[(a.to(DEV), e.to(DEV), s) for a, e, s in d.batch_adjs])
        logits = net(data)
        ref = u[tag + ".logits"]
        assert logits.dtype == torch.float32
        assert np.abs(logits.detach().cpu().numpy() - ref).max() <= 5e-2 * max(1.0, np.abs(ref).max() / 8), tag
        (logits * torch.from_numpy(g["G"]).to(DEV)).sum().backward()
        if tag == "plain":
            gmax = max(np.abs(u[k]).max() for k in u.files if k.startswith("plain.grad."))
            for k, p in net.named_parameters():
                r_ = torch.from_numpy(u["plain.grad." + k]).double().flatten()
                q = p.grad.double().cpu().flatten()
                if r_.abs().max() < 1e-3 * gmax:
                    continue
                cos = float((q @ r_) / (q.norm() * r_.norm()))
                assert cos > 0.99 and 0.9 < float(q.norm() / r_.norm()) < 1.1, (k, cos, float(q.norm() / r_.norm()))


@pytest.mark.parametrize("M,na,nb1,nb2,a_f32", [(5000, 128, 128, 128, 0), (333, 64, 28, 28, 0), (70000, 128, 64, 64, 0), (2048, 2, 64, 0, 1), (40000, 64, 20, 0, 0)])
def test_merged_bf16_weight_gradient_launch_matches_the_separate_calls(M, na, nb1, nb2, a_f32):
    """dgnn_linear_wgrad_bf16_cat: dW1 / dW2 bit for bit dgnn_linear_wgrad_bf16 (same row splits, same products), the bias sums against fp64"""
    from dgnn_amd import ops
    from dgnn_amd._lib import lib
    g = torch.Generator().manual_seed(M + nb2)
    A = torch.randn(M, na, generator=g).to(DEV)
    A = A if a_f32 else A.to(torch.bfloat16)
    B1 = torch.randn(M, nb1 + 8, generator=g).to(DEV).to(torch.bfloat16)[:, 8:]      # a strided view
    B2 = torch.randn(M, nb2, generator=g).to(DEV).to(torch.bfloat16) if nb2 else None
    dW1, dW2, db = (torch.empty(na, nb1, device=DEV), torch.empty(na, max(nb2, 1), device=DEV), torch.empty(na, device=DEV))
    scratch = torch.empty(int(lib().dgnn_linear_wgrad_cat_scratch_elems(M, na, nb1, nb2)), device=DEV)
    ops.check(lib().dgnn_linear_wgrad_bf16_cat(ops.ptr(A), a_f32, A.stride(0), na, ops.ptr(B1), B1.stride(0), nb1, ops.ptr(B2), B2.stride(0) if nb2 else 0, nb2, 0, M,
                                               ops.ptr(dW1), ops.ptr(dW2) if nb2 else None, ops.ptr(db), ops.ptr(scratch), ops.stream_ptr()), "dgnn_linear_wgrad_bf16_cat")
    assert torch.equal(dW1, ops.linear_wgrad(A, B1))
    if nb2:
        assert torch.equal(dW2, ops.linear_wgrad(A, B2))
    Ad = A.double()
    assert (db.double() - Ad.sum(0)).abs().max().item() <= 1e-6 * Ad.abs().sum(0).max().item()
    assert (dW1.double() - (A.to(torch.bfloat16).double().t() @ B1.double())).abs().max().item() <= 1e-4 * (A.double().abs().t() @ B1.double().abs()).max().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c_in,n_src,n_dst", [(64, 900, 600), (128, 5000, 1777), (32, 333, 333)])
def test_given_phi_aggregate_backward_with_the_two_additions(dtype, c_in, n_src, n_dst):
    """dgnn_sage_aggregate_bwd_phi_add against the launch chain it replaces (aggregate backward, then dx[:n] += add, then dphi += dphi_ext): fp32 bit for
    bit; bf16 storage rounds the sums once instead of twice (within one bf16 ulp of the two-step result)"""
    from dgnn_amd import ops
    from dgnn_amd._lib import lib
    if os.environ.get("DGNN_AGG_CHUNKED") == "0":
        pytest.skip("the addend form lives in the chunked kernel")
    g = torch.Generator().manual_seed(c_in + n_src)
    E = 4 * n_dst
    ei = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.arange(n_dst).repeat_interleave(4)])
    ei[0, : E // 3] = ei[0, : E // 3] % 50
    mk = lambda *sh: torch.randn(*sh, generator=g).to(DEV).to(dtype)
    x, da_w, phi, ext = mk(n_src, c_in), mk(n_dst, 2 * c_in), mk(E, c_in), mk(E, c_in)
    rowptr, _, _ = ops.plan_build(ei.to(DEV), n_dst, 1)
    t_rowptr, t_dst, t_eid = ops.plan_build(ei.to(DEV), n_src, 0)
    da, add = da_w[:, :c_in], da_w[:, c_in:]
    dx0, _, _, dphi0 = ops.aggregate_bwd(t_rowptr, t_dst, t_eid, n_src, rowptr, x, da.contiguous(), phi=phi)
    dx1, dphi1 = torch.empty_like(dx0), torch.empty_like(dphi0)
    ops.check(lib().dgnn_sage_aggregate_bwd_phi_add(ops.ptr(t_rowptr), ops.ptr(t_dst), ops.ptr(t_eid), n_src, ops.ptr(rowptr), ops.ptr(x), c_in, c_in, ops.ptr(phi), c_in,
                                                    ops.ptr(da), 2 * c_in, ops.ptr(dx1), c_in, ops.ptr(add), 2 * c_in, n_dst, ops.ptr(dphi1), c_in, ops.ptr(ext),
                                                    int(dtype == torch.bfloat16), ops.stream_ptr()), "dgnn_sage_aggregate_bwd_phi_add")
    if dtype == torch.float32:
        want_dx = dx0.clone()
        want_dx[:n_dst] += add
        assert torch.equal(dx1, want_dx) and torch.equal(dphi1, dphi0 + ext)
    else:
        want_dx = dx0.float()
        want_dx[:n_dst] += add.float()
        assert (dx1.float() - want_dx).abs().max().item() <= 2 ** -7 * want_dx.abs().max().item()
        want = dphi0.float() + ext.float()
        assert (dphi1.float() - want).abs().max().item() <= 2 ** -7 * want.abs().max().item()


@pytest.mark.parametrize("M,k1,k2,n_out,f32out", [(5000, 128, 128, 128, False), (2048, 64, 0, 2, True), (333, 20, 0, 64, False), (16384, 128, 0, 64, False), (77, 30, 28, 70, False)])
def test_small_bf16_gemm_gives_the_bits_of_the_tiled_kernel(M, k1, k2, n_out, f32out):
    """M <= 16384 takes the one-wavefront-per-block kernel: same k order per output element as the 128-row tiles, so the first M rows of a 20000-row
    problem (tiled kernel) carry the same bits; and both agree with fp64 at bf16-product accuracy"""
    from dgnn_amd import ops
    g = torch.Generator().manual_seed(M + k2)
    big = 20000
    A1 = torch.randn(big, k1 + 8, generator=g).to(DEV).to(torch.bfloat16)[:, 8:] if k1 % 8 == 0 else torch.randn(big, k1, generator=g).to(DEV).to(torch.bfloat16)
    W1 = torch.randn(n_out, k1, generator=g).to(DEV)
    A2 = torch.randn(big, k2, generator=g).to(DEV).to(torch.bfloat16) if k2 else None
    W2 = torch.randn(n_out, k2, generator=g).to(DEV) if k2 else None
    bias = torch.randn(n_out, generator=g).to(DEV)
    od = torch.float32 if f32out else None
    ref = ops.linear_fwd(A1, W1, A2, W2, bias, relu=True, out_dtype=od)
    got = ops.linear_fwd(A1[:M], W1, A2[:M] if k2 else None, W2, bias, relu=True, out_dtype=od)
    assert torch.equal(got, ref[:M])
    want = A1[:M].double() @ W1.to(torch.bfloat16).double().t() + bias.double()
    if k2:
        want = want + A2[:M].double() @ W2.to(torch.bfloat16).double().t()
    want = torch.relu(want)
    assert (got.double() - want).abs().max().item() <= (2 ** -7 if not f32out else 1e-4) * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c_in,n_src,n_dst", [(28, 5000, 1777), (64, 900, 600), (128, 3000, 2000), (32, 70, 70)])
def test_lane_group_given_phi_backward_gives_the_bits_of_the_lane_per_channel_kernel(dtype, c_in, n_src, n_dst):
    """k_agg_bwd_g (4 channels per lane, 64 / G source rows per wavefront) against k_agg_bwd_c (taken when a row stride is not a multiple of 4
    elements): dx and dphi bit for bit; sources with dozens of out-edges (chunks beyond 64 edges), with none, degree-0 destinations' clamp"""
    from dgnn_amd import ops
    if os.environ.get("DGNN_AGG_CHUNKED") == "0" or os.environ.get("DGNN_AGG_GROUPED") == "0":
        pytest.skip("compares the two default kernels")
    g = torch.Generator().manual_seed(c_in + n_src)
    E = 4 * n_dst
    ei = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.arange(n_dst).repeat_interleave(4)])
    ei[0, : E // 3] = ei[0, : E // 3] % 37          # 37 sources with dozens of out-edges each
    mk = lambda *sh: torch.randn(*sh, generator=g).to(DEV).to(dtype)
    x, da, phi = mk(n_src, c_in), mk(n_dst, c_in), mk(E, c_in)
    rowptr, _, _ = ops.plan_build(ei.to(DEV), n_dst, 1)
    t_rowptr, t_dst, t_eid = ops.plan_build(ei.to(DEV), n_src, 0)
    dx_g, _, _, dphi_g = ops.aggregate_bwd(t_rowptr, t_dst, t_eid, n_src, rowptr, x, da, phi=phi)
    pad = 2 if dtype == torch.bfloat16 else 1        # even element counts keep bf16 pairs aligned; the stride is no multiple of 4 either way
    xw = torch.zeros(n_src, c_in + pad, device=DEV, dtype=dtype)
    xw[:, :c_in] = x
    dx_c, _, _, dphi_c = ops.aggregate_bwd(t_rowptr, t_dst, t_eid, n_src, rowptr, xw[:, :c_in], da, phi=phi)
    assert torch.equal(dx_g, dx_c) and torch.equal(dphi_g, dphi_c)
    # the forward pair (k_agg_fwd_g vs k_agg_fwd), on a plan with ragged in-degrees (0 .. dozens) as well
    ei2 = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, n_dst, (E,), generator=g)])
    ei2[1, : E // 4] = ei2[1, : E // 4] % 11
    for edges in (ei, ei2):
        rp2, src2, eid2 = ops.plan_build(edges.to(DEV), n_dst, 1)
        a_g = ops.aggregate_fwd(rp2, src2, eid2, n_dst, x, phi=phi)
        a_c = ops.aggregate_fwd(rp2, src2, eid2, n_dst, xw[:, :c_in], phi=phi)
        assert torch.equal(a_g, a_c)
    # and against the definition in fp64 (on the storage-rounded inputs)
    cnt = torch.bincount(ei[1], minlength=n_dst).clamp_min(1).to(DEV).double()
    dm = da.double()[ei[1].to(DEV)] / cnt[ei[1].to(DEV)][:, None]
    want = torch.zeros(n_src, c_in, device=DEV, dtype=torch.float64).index_add_(0, ei[0].to(DEV), dm * phi.double())
    tol = 2 ** -7 if dtype == torch.bfloat16 else 1e-5
    assert (dx_g.double() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
