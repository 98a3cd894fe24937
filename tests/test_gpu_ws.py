"""The wave-specialised fused layers (csrc/fused_ws.hip, round 5) on their own: SAGEConv.forward + BatchNorm(eval) + ReLU of the reference
(learning/surfaceNetStaticEdgeFilters.py:66-96, :345-346) for the shapes the kernel takes -- 64 -> 128 and 128 -> 128, fp32 rows and the unsigned 16-bit rows
of the bf16-storage chain, the decoder inside the last launch (:180-187, :350-351) -- against fp64 on the same inputs, on graphs that exercise what the
hand-off protocol and the tiling can get wrong: cell counts that leave 1 / 9 / 31 cells (or whole empty producer groups) in the last tile, fewer tiles than
workgroups, ragged in-degrees (the per-lane path inside a tile of matrix-core groups), padded row strides, destination sub-ranges, edge rows through `eid`;
and repeated launches bit for bit.  The whole-model parity suites (tests/test_gpu_parity.py, test_gpu_infer.py, test_gpu_bf16.py) run the same kernels
through the module interface; tests/test_gpu_infer.py::test_ignatius_layers_repeat_bit_for_bit_with_cold_caches pins the race the hand-off once had."""
import os

import numpy as np
import pytest
import torch

from test_gpu_parity import DEV

pytestmark = pytest.mark.gpu
BOUND_SCALE = float(os.environ.get("DGNN_TEST_BOUND_SCALE", "1"))      # (< 1: how much margin the stated bounds have on this box)


def _graph(n, seed, ragged):
    """edge_index int64 [2, E]: 4-regular in the reference layout (4 rows per destination), or ragged (in-degrees 0 .. 9, every fourth group regular)"""
    rng = np.random.default_rng(seed)
    if not ragged:
        return np.stack([rng.integers(0, n, 4 * n), np.repeat(np.arange(n), 4)]).astype(np.int64)
    deg = rng.integers(0, 10, n)
    keep = (np.arange(n) // 4) % 4 == 0          # groups of four cells the kernel takes on the matrix cores, between groups it does not
    deg[keep] = 4
    dst = np.repeat(np.arange(n), deg)
    return np.stack([rng.integers(0, n, dst.shape[0]), dst]).astype(np.int64)


def _layer_inputs(c_in, n, E, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, c_in, generator=g))
    x[::37] *= 50.0                                # rows far apart in magnitude: the per-row power-of-two scales
    ea = torch.randn(E, 20, generator=g)
    ea[::29] *= 20.0
    We, be = torch.randn(c_in, 20, generator=g) * 0.3, torch.randn(c_in, generator=g)
    Wj, Wi, bj = torch.randn(128, c_in, generator=g) * 0.1, torch.randn(128, c_in, generator=g) * 0.1, torch.randn(128, generator=g)
    Wj[::7] *= 30.0                                # weight rows far apart: the per-consumer scales
    scale, shift = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    return x, ea, We, be, Wj, bj, Wi, scale, shift


def _reference(x, ea, ei, n, We, be, Wj, bj, Wi, scale, shift):
    """fp64: (result, magnitude of the terms that were added up)"""
    from oracle.pyg_semantics import propagate_mean
    d = lambda t: t.double()
    a = propagate_mean(d(x), n, ei, d(ea) @ d(We).t() + d(be))
    amag = propagate_mean(d(x).abs(), n, ei, d(ea).abs() @ d(We).abs().t() + d(be).abs())
    ref = torch.relu((a @ d(Wj).t() + d(x)[:n] @ d(Wi).t() + d(bj)) * d(scale) + d(shift))
    mag = (amag @ d(Wj).abs().t() + d(x)[:n].abs() @ d(Wi).abs().t() + d(bj).abs()) * d(scale).abs() + d(shift).abs()
    return ref, mag


@pytest.mark.parametrize("c_in", [64, 128])
@pytest.mark.parametrize("n,ragged", [(1, False), (9, False), (31, False), (33, False), (32 * 7 + 9, False), (32 * 300 + 1, False), (32 * 300 + 13, True), (2500, True)])
def test_ws_layer_vs_fp64(c_in, n, ragged):
    from dgnn_amd import ops
    if ops.GEMM_MODE != ops.GEMM_F16X2 or not ops.lib().dgnn_wave_specialised_enabled():
        pytest.skip("the wave-specialised kernel runs the default arithmetic (DGNN_GEMM_MODE / DGNN_WS select the two-phase kernels)")
    ei = torch.from_numpy(_graph(n, 100 * c_in + n, ragged))
    E = ei.size(1)
    x, ea, We, be, Wj, bj, Wi, scale, shift = _layer_inputs(c_in, n, E, n + c_in)
    ref, mag = _reference(x, ea, ei, n, We, be, Wj, bj, Wi, scale, shift)
    dv = lambda *ts: [t.to(DEV) for t in ts]
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n, 1, n_other=n)
    xs = torch.zeros(n, c_in + 12, device=DEV)[:, :c_in]                 # padded row stride (a multiple of 4 floats, rows 16-byte aligned)
    xs.copy_(x)
    out = ops.sage_layer_fused_fwd(rowptr, src, n, xs, ea.to(DEV), *dv(We, be, Wj, bj, Wi, scale, shift), True, eid=eid)
    err = (out.cpu().double() - ref).abs()
    # 22 significand bits per operand, fp32 accumulation of up to 2 c_in + 84 terms: a few 2^-22 of the terms' magnitude
    bound = BOUND_SCALE * 2.0 ** -19 * mag + 1e-30
    assert bool((err <= bound).all()), (float((err / bound).max()), int((err > bound).sum()))
    for _ in range(3):
        assert torch.equal(out, ops.sage_layer_fused_fwd(rowptr, src, n, xs, ea.to(DEV), *dv(We, be, Wj, bj, Wi, scale, shift), True, eid=eid))
    # edge rows already in plan order (eid = None), contiguous rows: the same bits
    ea_p = ea.to(DEV)[eid.long()]
    assert torch.equal(out, ops.sage_layer_fused_fwd(rowptr, src, n, x.to(DEV), ea_p, *dv(We, be, Wj, bj, Wi, scale, shift), True))
    # destination sub-ranges (interior / boundary launches of a partition): every cell's row bit for bit
    if n >= 40:
        cut = [0, 4 * (n // 12), 4 * (n // 12) + 8, n]      # (multiples of 4: a cell's group of four -- matrix-core or per-lane path -- is the whole launch's)
        parts = torch.full_like(out, float("nan"))
        for b, e in zip(cut[:-1], cut[1:]):
            ops.sage_layer_fused_fwd(rowptr[b:e + 1], src, e - b, xs, ea.to(DEV), *dv(We, be, Wj, bj, Wi, scale, shift), True, eid=eid, x_dst=xs[b:e], out=parts[b:e])
        assert torch.equal(parts, out)


@pytest.mark.parametrize("c_in", [64, 128])
@pytest.mark.parametrize("n,ragged", [(9, False), (32 * 200 + 1, False), (32 * 150 + 13, True)])
def test_ws_layer_on_unsigned_16_bit_rows_vs_fp64(c_in, n, ragged):
    from dgnn_amd import ops
    from test_gpu_bf16 import EPS, _ub_decode, _ub_round
    if ops.BF16_MODE != ops.BF16_COMPENSATED or not ops.lib().dgnn_wave_specialised_enabled():
        pytest.skip("unsigned rows belong to the compensated arithmetic; DGNN_WS=0 keeps the two-phase kernels")
    ei = torch.from_numpy(_graph(n, 300 * c_in + n, ragged))
    x, ea, We, be, Wj, bj, Wi, scale, shift = _layer_inputs(c_in, n, ei.size(1), 3 * n + c_in)
    xr = _ub_round(x)
    xin = ((xr.view(torch.int32) >> 15).to(torch.int16)).to(DEV)       # the rows as a previous layer stores them
    assert torch.equal(_ub_decode(xin), xr.double())
    ref, mag = _reference(xr, ea, ei, n, We, be, Wj, bj, Wi, scale, shift)
    dv = lambda *ts: [t.to(DEV) for t in ts]
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n, 1, n_other=n)
    out = ops.sage_layer_fused_fwd_bf16(rowptr, src, n, xin, c_in, ea.to(DEV), *dv(We, be, Wj, bj, Wi, scale, shift), True, eid=eid, rows_out_unsigned=True)
    assert out.dtype == ops.UROWS and out.shape == (n, 128)
    err = (_ub_decode(out) - ref).abs()
    # the stored value is off by at most half of the format's spacing (9 significant bits: 2^-9 of the value; EPS = 2^-8) + the arithmetic's few 2^-22 of the terms
    bound = 0.5 * EPS * ref.abs() + BOUND_SCALE * 2.0 ** -19 * mag + 1e-30
    assert bool((err <= bound).all()), (float((err / bound).max()), int((err > bound).sum()))
    for _ in range(3):
        assert torch.equal(out, ops.sage_layer_fused_fwd_bf16(rowptr, src, n, xin, c_in, ea.to(DEV), *dv(We, be, Wj, bj, Wi, scale, shift), True, eid=eid,
                                                              rows_out_unsigned=True))


@pytest.mark.parametrize("n,ragged", [(5, False), (32 * 3 + 9, False), (32 * 257 + 9, False), (32 * 257 + 21, True)])
def test_ws_last_layer_with_the_decoder_vs_fp64(n, ragged):
    """the decoder-carrying launch: stage A / B / C one tile apart, stage B split between the roles -- the drain of the last two tiles, workgroups with 0 / 1 / 2
    tiles, and the logits of every cell against fp64; repeats and destination sub-ranges bit for bit"""
    from dgnn_amd import ops
    if ops.GEMM_MODE != ops.GEMM_F16X2 or not ops.lib().dgnn_wave_specialised_enabled() or not ops.FUSE_DECODER:
        pytest.skip("the decoder-carrying wave-specialised launch is the default arithmetic's (DGNN_GEMM_MODE / DGNN_WS / DGNN_FUSE_DECODER)")
    ei = torch.from_numpy(_graph(n, 7000 + n, ragged))
    x, ea, We, be, Wj, bj, Wi, scale, shift = _layer_inputs(128, n, ei.size(1), 5 * n)
    g = torch.Generator().manual_seed(n)
    W0, b0 = torch.randn(64, 128, generator=g) * 0.15, torch.randn(64, generator=g)
    s1, h1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    W3, b3 = torch.randn(2, 64, generator=g) * 0.3, torch.randn(2, generator=g)
    d = lambda t: t.double()
    ref, mag = _reference(x, ea, ei, n, We, be, Wj, bj, Wi, scale, shift)
    hid = torch.relu((ref @ d(W0).t() + d(b0)) * d(s1) + d(h1))
    lref = hid @ d(W3).t() + d(b3)
    lmag = (((mag @ d(W0).abs().t() + d(b0).abs()) * d(s1).abs() + d(h1).abs()) @ d(W3).abs().t()) + d(b3).abs()
    dv = lambda *ts: [t.to(DEV) for t in ts]
    rowptr, src, eid = ops.plan_build(ei.to(DEV), n, 1, n_other=n)
    args = dv(We, be, Wj, bj, Wi, scale, shift)
    lg = ops.sage_layer_fused_decoder_fwd(rowptr, src, n, x.to(DEV), ea.to(DEV), *args, True, *dv(W0, b0, s1, h1, W3, b3), eid=eid)
    err = (lg.cpu().double() - lref).abs()
    bound = BOUND_SCALE * 2.0 ** -18 * lmag + 1e-30
    assert bool((err <= bound).all()), (float((err / bound).max()), int((err > bound).sum()))
    for _ in range(5):
        assert torch.equal(lg, ops.sage_layer_fused_decoder_fwd(rowptr, src, n, x.to(DEV), ea.to(DEV), *args, True, *dv(W0, b0, s1, h1, W3, b3), eid=eid))
