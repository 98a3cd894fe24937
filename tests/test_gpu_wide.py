"""Wide conv layers on SPLIT ROWS (csrc/wide.hip; include/dgnn_hip.h "SPLIT ROWS"): the storage format against a numpy restatement of its definition
(bit for bit), the dense product and the aggregate against fp64, the whole model at the reference's wide widths (configs/eth.yaml:56, aerial.yaml:57:
[64,128,256,512]; configs/modelnet.yaml:56: [128,256,512,1024]) against the CPU oracle (reference learning/surfaceNetStaticEdgeFilters.py:66-96,
:323-355) and against the fp32-row path of rounds 2-4."""
import numpy as np
import pytest
import torch

from dgnn_amd.config import Config
from helpers import oracle_static
from test_gpu_parity import DEV, TOL_LOGIT, hip_static

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _default_arithmetic_only():
    """split rows are the default arithmetic's form: under DGNN_GEMM_MODE=<other> the models take the fp32-row path at these widths and this suite has
    nothing to test"""
    from dgnn_amd import ops
    if ops.GEMM_MODE != ops.GEMM_F16X2:
        pytest.skip("split rows exist for the default arithmetic (f16x2) only")


def PI(p):
    return (p & 3) | ((p >> 4) << 2) | (((p >> 2) & 3) << 3)


def np_pack(x, gch):
    """the format's definition in numpy: x fp32 [n, C] -> (bytes uint8 [n, C * 4], scales fp32 [n, ng])"""
    n, C = x.shape
    nch = C // 32
    g = gch if gch > 0 else nch
    ng = (nch + g - 1) // g
    data = np.zeros((n, nch, 2, 32), dtype=np.float16)
    scales = np.zeros((n, ng), dtype=np.float32)
    perm = np.array([PI(p) for p in range(32)])
    for gi in range(ng):
        c0, c1 = gi * g * 32, min(C, (gi + 1) * g * 32)
        m = np.abs(x[:, c0:c1]).max(axis=1).astype(np.float32)
        E = (m.view(np.uint32) >> 23).astype(np.int64)
        zero = E <= 14
        Ec = np.minimum(E, 254)
        s = ((268 - Ec).astype(np.uint32) << 23).view(np.float32)
        s_store = np.where(zero, np.float32(2.0 ** 127), s)
        s_mul = np.where(zero, np.float32(0), s)
        scales[:, gi] = s_store
        for q in range(c0 // 32, c1 // 32):
            v = (x[:, q * 32:(q + 1) * 32][:, perm] * s_mul[:, None]).astype(np.float32)
            with np.errstate(over="ignore", invalid="ignore"):
                hi = v.astype(np.float16)
                lo = (v - hi.astype(np.float32)).astype(np.float16)
            data[:, q, 0], data[:, q, 1] = hi, lo
    return data.reshape(n, -1).view(np.uint8), scales


def wide_rows(n, C, seed, spread=True):
    """fp32 rows whose rows span 40 binades and whose 256-channel groups differ by up to 2^12; some all-zero groups, one dropped group, tiny values"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, C, generator=g)
    if spread:
        x = x * torch.exp2(torch.randint(-20, 21, (n, 1), generator=g).float())
        for gi in range((C + 255) // 256):
            x[:, gi * 256:(gi + 1) * 256] *= torch.exp2(torch.randint(-6, 7, (n, 1), generator=g).float())
        if n > 8:
            x[3, :min(C, 256)] = 0.0                       # an all-zero group
            x[5] = 0.0                                     # an all-zero row
            x[7, :min(C, 256)] *= 2.0 ** -60               # a group far below the rest of its row (dropped by the product when C > 256)
            x[9, 1] = 1e-39                                # a subnormal among normal values
    return x


@pytest.mark.parametrize("C,per_row", [(128, False), (256, False), (512, False), (1024, False), (512, True), (64, True)])
def test_split_rows_are_the_stated_format(C, per_row):
    from dgnn_amd import ops
    x = wide_rows(300, C, C + int(per_row))
    sr = ops.pack_rows(x.to(DEV), per_row=per_row)
    want_b, want_s = np_pack(x.numpy(), 0 if per_row else 8)
    assert sr.data.shape == (300, C * 4) and np.array_equal(sr.scales.cpu().numpy(), want_s)
    assert np.array_equal(sr.data.cpu().numpy(), want_b)
    # and back: hi + lo carries 22 significand bits of every value that is not far below its group's largest
    back = sr.float().cpu()
    gmax = torch.stack([x[:, g * 256:(g + 1) * 256].abs().max(1).values for g in range((C + 255) // 256)], 1).repeat_interleave(256, 1)[:, :C] if not per_row \
        else x.abs().max(1, keepdim=True).values.expand(-1, C)
    live = gmax > 2.0 ** -112
    err = (back - x).abs()
    assert (err[live] <= 2.0 ** -21 * x.abs()[live] + 2.0 ** -38 * gmax[live]).all() and (back[~live] == 0).all()
    two = ops.pack_rows(x[:, :C // 2].contiguous().to(DEV), x[:, C // 2:].contiguous().to(DEV), per_row=per_row)
    assert torch.equal(two.data, sr.data) and torch.equal(two.scales, sr.scales)       # [A1 | A2] packs like the concatenated row


def _ref_linear(a1, a2, W1, W2, bias, scale, shift, relu):
    ref = a1.double() @ W1.double().t() + bias.double()
    mag = a1.double().abs() @ W1.double().abs().t()
    if a2 is not None:
        ref = ref + a2.double() @ W2.double().t()
        mag = mag + a2.double().abs() @ W2.double().abs().t()
    ref = ref * scale.double() + shift.double()
    mag = mag * scale.double().abs() + bias.double().abs() * scale.double().abs() + shift.double().abs()
    return (ref.clamp_min(0) if relu else ref), mag


@pytest.mark.parametrize("c1,c2,n_out,M", [(256, 256, 512, 1000), (128, 128, 256, 517), (512, 512, 1024, 300), (512, 0, 256, 777), (1024, 0, 512, 260),
                                            (256, 256, 512, 64), (128, 128, 256, 31), (512, 0, 256, 1)])
def test_linear_sr_vs_fp64(c1, c2, n_out, M):
    """the dense product on split rows: both operand parts with their own group scales (rows over 40 binades, groups 2^12 apart, zero groups, a group
    2^60 below its row), weight rows over 20 binades, BatchNorm / ReLU epilogue, a ragged last tile -- against fp64 on the operands' STORED values;
    split-row output == fp32 output repacked, bit for bit; a prefix of the rows gives the same bits (a cell's result depends on its own row only)"""
    from dgnn_amd import ops
    g = torch.Generator().manual_seed(c1 + 3 * n_out)
    a1 = ops.pack_rows(wide_rows(M, c1, 1).to(DEV))
    a2 = ops.pack_rows(wide_rows(M, c2, 2).to(DEV)) if c2 else None
    W = torch.randn(n_out, c1 + c2, generator=g) / (c1 + c2) ** 0.5 * torch.exp2(torch.randint(-10, 11, (n_out, 1), generator=g).float())
    bias, scale, shift = torch.randn(n_out, generator=g), torch.rand(n_out, generator=g) + 0.5, torch.randn(n_out, generator=g) * 0.1
    W1, W2 = W[:, :c1].contiguous().to(DEV), (W[:, c1:].contiguous().to(DEV) if c2 else None)
    Wp = ops.pack_rows(W1, W2, per_row=True)
    Ws = Wp.float().cpu()          # the weights as stored (22 bits)
    for relu in (True, False):
        o32 = ops.linear_sr(a1, Wp, a2, bias.to(DEV), scale.to(DEV), shift.to(DEV), relu=relu, out_f32=True)
        osr = ops.linear_sr(a1, Wp, a2, bias.to(DEV), scale.to(DEV), shift.to(DEV), relu=relu)
        assert o32 is not None and osr is not None and o32.shape == (M, n_out) and osr.channels == n_out
        ref, mag = _ref_linear(a1.float().cpu(), a2.float().cpu() if a2 is not None else None, Ws[:, :c1], Ws[:, c1:] if c2 else None, bias, scale, shift, relu)
        err = (o32.double().cpu() - ref).abs()
        assert torch.isfinite(o32).all() and (err <= 2e-6 * mag + 1e-30).all(), (err / (mag + 1e-300)).max().item()
        again = ops.pack_rows(o32)
        assert torch.equal(again.data, osr.data) and torch.equal(again.scales, osr.scales)
        k = M - 131 if M > 131 else max(1, M // 2)
        part = ops.linear_sr(a1, Wp, a2, bias.to(DEV), scale.to(DEV), shift.to(DEV), relu=relu, out_f32=True, rows=k)
        assert torch.equal(part, o32[:k])
        # the decoder's output Linear inside the launch (reference :180-187): logits = act(...) . W3^T + b3, the hidden rows never stored
        for n_proj in (2, 1):
            W3 = torch.randn(n_proj, n_out, generator=g) / n_out ** 0.5
            b3 = torch.randn(n_proj, generator=g)
            lg = ops.linear_sr(a1, Wp, a2, bias.to(DEV), scale.to(DEV), shift.to(DEV), relu=relu, proj=(W3.to(DEV), b3.to(DEV)))
            if n_out > 512:
                assert lg is None          # more than two column tiles would make the atomic sum order-dependent: declined
                continue
            want = o32.double().cpu() @ W3.double().t() + b3.double()
            wmag = o32.double().cpu().abs() @ W3.double().abs().t() + b3.double().abs()
            assert lg.shape == (M, n_proj) and ((lg.double().cpu() - want).abs() <= 2e-6 * wmag + 1e-30).all()
            assert torch.equal(lg, ops.linear_sr(a1, Wp, a2, bias.to(DEV), scale.to(DEV), shift.to(DEV), relu=relu, proj=(W3.to(DEV), b3.to(DEV))))


def _graph(n, seed, ragged):
    """(edge_index int64 [2, E]) 4-regular in the reference layout, or ragged (in-degrees 0 .. 9)"""
    rng = np.random.default_rng(seed)
    if not ragged:
        dst = np.stack([rng.permutation(n) for _ in range(4)], 1).reshape(-1)         # every cell 4 out-edges; in-degree 4 as well (4 permutations)
        return np.stack([np.repeat(np.arange(n), 4), dst]).astype(np.int64)
    deg = rng.integers(0, 10, n)
    dst = np.repeat(np.arange(n), deg)
    return np.stack([rng.integers(0, n, dst.shape[0]), dst]).astype(np.int64)


@pytest.mark.parametrize("C,sr_in,ragged", [(128, False, False), (256, True, False), (512, True, False), (128, False, True), (256, True, True), (512, True, True)])
def test_aggregate_sr_vs_fp64(C, sr_in, ragged):
    """a = mean_j x_j * lin_e(edge_attr_j) (reference :75-80, :89-96; PyG propagate + scatter(mean): oracle/pyg_semantics.py) as split rows, from fp32
    source rows (the layer behind a fused layer; its own rows come back as split rows too) and from split rows; 4-regular groups on the matrix cores,
    other in-degrees on the per-lane path; attributes over 30 binades; n_dst a prefix of the sources; a destination prefix gives the same bits"""
    from dgnn_amd import ops
    from dgnn_amd.graph import GraphPlan
    n = 1003
    ei = _graph(n, C + ragged, ragged)
    g = torch.Generator().manual_seed(C)
    x = wide_rows(n, C, 4)
    E = ei.shape[1]
    ea = torch.randn(E, 20, generator=g) * torch.exp2(torch.randint(-15, 16, (E, 1), generator=g).float())
    We = torch.randn(C, 20, generator=g) / 20 ** 0.5
    be = torch.randn(C, generator=g)
    eit = torch.from_numpy(ei).to(DEV)
    plan = GraphPlan(eit, n, n)
    xin = ops.pack_rows(x.to(DEV)) if sr_in else x.to(DEV)
    xst = xin.float().cpu() if sr_in else x            # the source rows as stored
    prep = ops.sr_prepare_filter(We.to(DEV), be.to(DEV))
    got = ops.aggregate_sr(plan.rowptr, plan.src, plan.eid, n, xin, ea.to(DEV), We.to(DEV), be.to(DEV), prep, own_rows=not sr_in)
    a, own = got if not sr_in else (got, None)
    phi = ea.double() @ We.double().t() + be.double()
    m = xst.double()[ei[0]] * phi
    S = torch.zeros(n, C, dtype=torch.float64).index_add_(0, torch.from_numpy(ei[1]), m)
    Sm = torch.zeros(n, C, dtype=torch.float64).index_add_(0, torch.from_numpy(ei[1]), xst.double().abs()[ei[0]] * (ea.double().abs() @ We.double().abs().t() + be.double().abs()))
    cnt = torch.bincount(torch.from_numpy(ei[1]), minlength=n).clamp(min=1).double()[:, None]
    ref, mag = S / cnt, Sm / cnt
    back = a.float().cpu().double()
    gmax = torch.stack([ref[:, gg * 256:(gg + 1) * 256].abs().max(1).values for gg in range((C + 255) // 256)], 1).repeat_interleave(256, 1)[:, :C]
    err = (back - ref).abs()
    assert torch.isfinite(back).all() and (err <= 1e-6 * mag + 2.0 ** -21 * gmax + 1e-30).all(), (err / (mag + gmax * 2.0 ** -21 + 1e-300)).max().item()
    if own is not None:
        want = ops.pack_rows(x.to(DEV))
        assert torch.equal(own.data, want.data) and torch.equal(own.scales, want.scales)
    # SMALL destination sets (the inner blocks of a mini-batched schedule, reference :232-275): fewer groups of four than the launch has XCD eighths.
    # Round 6: below 256 destination cells the launch had fewer than 8 workgroups and the eighths without one were never computed.
    if not ragged:
        for k_small in (1, 4, 31, 64, 100, 152, 255):
            sm = ops.aggregate_sr(plan.rowptr[:k_small + 1], plan.src, plan.eid, k_small, xin, ea.to(DEV), We.to(DEV), be.to(DEV), prep)
            assert torch.equal(sm.data, a.data[:k_small]) and torch.equal(sm.scales, a.scales[:k_small]), k_small
    # a destination prefix of the same plan: same bits for its cells
    k = n - 402
    pre = ops.aggregate_sr(plan.rowptr[:k + 1], plan.src, plan.eid, k, xin, ea.to(DEV), We.to(DEV), be.to(DEV), prep)
    if not ragged:
        assert torch.equal(pre.data, a.data[:k]) and torch.equal(pre.scales, a.scales[:k])
    else:
        # a ragged graph: whether a cell's group of four takes the matrix-core path depends on its group mates (as in the fused layers): the cut-off
        # group agrees to rounding, every whole group before it bit for bit
        k4 = k // 4 * 4
        assert torch.equal(pre.data[:k4], a.data[:k4]) and (pre.float().cpu().double() - ref[:k]).abs().le(1e-6 * mag[:k] + 2.0 ** -21 * gmax[:k] + 1e-30).all()


@pytest.mark.parametrize("convs", [(64, 128, 256, 512), (128, 256, 512, 1024)])
def test_wide_widths_whole_model_vs_oracle_and_the_fp32_row_path(convs):
    """inference_layer at the widths the reference's real configs use: the wide layers (and the decoder's first Linear) run on split rows -- logits
    against the CPU oracle within the fp32 tolerance (1e-4), against the fp32-row path of rounds 2-4 (DGNN_WIDE_SR=0) at fp32 rounding level, and the
    same bits on a second call"""
    from dgnn_amd import ops
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, _, _ = delaunay_tet_graph(6000, 5)
    n = adj.shape[0] // 4
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    onet = oracle_static(convs=convs, load=False, seed=3)
    for m in onet.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    sd = onet.state_dict()
    net = hip_static(convs=convs, sd=sd)
    assert net.wide_layer_split_rows(convs[1], convs[2]) and net.wide_layer_split_rows(convs[2], convs[3])
    data = Config(x=x, edge_attr=ea, edge_index=ei)
    calls = []
    real = ops.linear_sr
    ops.linear_sr = lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1]
    try:
        got = net.inference_layer(data)
    finally:
        ops.linear_sr = real
    assert len(calls) == (3 if convs[0] == 64 else 4)          # the wide conv layers + the decoder (hidden layer and output Linear in one launch)
    with torch.no_grad():
        want = onet.inference_layer(Config(x=x.cpu(), edge_attr=ea.cpu(), edge_index=ei.cpu()))
    err = (got.cpu() - want).abs()
    assert err.max().item() <= TOL_LOGIT * max(1.0, want.abs().max().item()), err.max().item()
    assert torch.equal(net.inference_layer(data), got)
    old = ops.WIDE_SR
    ops.WIDE_SR = False
    try:
        base = net.inference_layer(data)
    finally:
        ops.WIDE_SR = old
    assert (base - got).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("convs", [(64, 128, 256, 512)])
def test_wide_widths_minibatch_schedules_vs_whole_graph(convs):
    """The two mini-batched inference schedules at the reference's wide widths (configs/eth.yaml:56 validates with per_layer=1 + batch_size:
    reference inference_layer_batch :279-320 via learning/runModel.py:405; inference_batch_layer :232-275): the wide layers hand back split rows
    (ops.SplitRows) and both schedules must still run and agree with whole-graph inference_layer within the fp32 tolerance."""
    from dgnn_amd.sampler import NeighborSampler
    from dgnn_amd.synthetic import delaunay_tet_graph, hashed_normal
    adj, _, _ = delaunay_tet_graph(900, 6)
    n = adj.shape[0] // 4
    x = hashed_normal(np.arange(n), 29, seed=1, device=DEV)
    ea = hashed_normal(np.arange(4 * n), 20, seed=2, device=DEV)
    ei = torch.from_numpy(adj.T.astype(np.int64)).to(DEV)
    onet = oracle_static(convs=convs, load=False, seed=4)
    for m in onet.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    net = hip_static(convs=convs, sd=onet.state_dict())
    assert net.wide_layer_split_rows(convs[1], convs[2]) and net.wide_layer_split_rows(convs[2], convs[3])
    full = net.inference_layer(Config(x=x, edge_attr=ea, edge_index=ei))
    tol = TOL_LOGIT * max(1.0, full.abs().max().item())
    one_hop = NeighborSampler(ei, sizes=[-1], num_nodes=n, batch_size=1024, shuffle=False)
    lb = net.inference_layer_batch(Config(x=x, edge_attr=ea), one_hop)
    assert lb.shape == full.shape and (lb - full).abs().max().item() <= tol
    sel = torch.arange(0, n, 7, device=DEV)
    k_hop = NeighborSampler(ei, sizes=[-1] * 4, node_idx=sel, num_nodes=n, batch_size=256, shuffle=False)
    bl = net.inference_batch_layer(Config(x=x, edge_attr=ea), k_hop)
    assert (bl[sel] - full[sel]).abs().max().item() <= tol
