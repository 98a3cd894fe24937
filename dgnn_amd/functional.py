"""Differentiable building blocks (torch.autograd.Function over the HIP kernels).

These replace what autograd records for the reference's op chain (learning/runModel.py:279 runs
``loss.backward()`` through lin_e -> index_select -> mul -> scatter-mean -> lin_j/lin_i -> BatchNorm
-> ReLU).  The filter phi is recomputed in the backward kernel instead of being saved, so a layer
saves x_src and the aggregate only -- not the three [E, C_in] tensors PyTorch keeps.
"""
from __future__ import annotations

import torch

from . import ops
from .graph import GraphPlan


class _Aggregate(torch.autograd.Function):
    """a = mean_{e -> i}( x_src[src_e] * (We.A_e + be) )    (fused filter)"""

    @staticmethod
    def forward(ctx, x_src, edge_attr, We, be, plan: GraphPlan):
        a = ops.aggregate_fwd(plan.rowptr, plan.src, plan.eid, plan.n_dst, x_src, edge_attr, We, be)
        ctx.plan = plan
        ctx.edge_index = plan.edge_index  # the lazily built transposed plan (backward) reads it
        ctx.save_for_backward(x_src, edge_attr, We, be)
        return a

    @staticmethod
    def backward(ctx, da):
        x_src, edge_attr, We, be = ctx.saved_tensors
        plan = ctx.plan
        t_rowptr, t_dst, t_eid = plan.transposed
        da = da.contiguous()
        dx, dWe, dbe, _ = ops.aggregate_bwd(t_rowptr, t_dst, t_eid, plan.n_src, plan.rowptr, x_src, da, edge_attr, We, be,
                                            need_dx=ctx.needs_input_grad[0])
        return dx, None, dWe, dbe, None


class _AggregatePhi(torch.autograd.Function):
    """a = mean_{e -> i}( x_src[src_e] * phi_e ), phi given per edge row (Updated variant, edge MLPs)"""

    @staticmethod
    def forward(ctx, x_src, phi, plan: GraphPlan):
        a = ops.aggregate_fwd(plan.rowptr, plan.src, plan.eid, plan.n_dst, x_src, phi=phi)
        ctx.plan = plan
        ctx.edge_index = plan.edge_index  # the lazily built transposed plan (backward) reads it
        ctx.save_for_backward(x_src, phi)
        return a

    @staticmethod
    def backward(ctx, da):
        x_src, phi = ctx.saved_tensors
        plan = ctx.plan
        t_rowptr, t_dst, t_eid = plan.transposed
        dx, _, _, dphi = ops.aggregate_bwd(t_rowptr, t_dst, t_eid, plan.n_src, plan.rowptr, x_src, da.contiguous(), phi=phi,
                                           need_dx=ctx.needs_input_grad[0])
        return dx, dphi, None


class _AggregatePlain(torch.autograd.Function):
    """a = mean_{e -> i} x_src[src_e]   (lin_e is None: model.edge_convs == 0)"""

    @staticmethod
    def forward(ctx, x_src, plan: GraphPlan):
        ctx.plan = plan
        ctx.edge_index = plan.edge_index  # the lazily built transposed plan (backward) reads it
        ctx.save_for_backward(x_src)
        return ops.aggregate_fwd(plan.rowptr, plan.src, plan.eid, plan.n_dst, x_src)

    @staticmethod
    def backward(ctx, da):
        (x_src,) = ctx.saved_tensors
        plan = ctx.plan
        t_rowptr, t_dst, t_eid = plan.transposed
        dx, _, _, _ = ops.aggregate_bwd(t_rowptr, t_dst, t_eid, plan.n_src, plan.rowptr, x_src, da.contiguous())
        return dx, None


class _Linear2(torch.autograd.Function):
    """out = A1.W1^T + A2.W2^T + bias   (A2/W2/bias optional).  bf16 storage: A1/A2 bf16, weights fp32 (master copies), out
    bf16 or -- `out_f32`, the logits -- fp32; gradients of activations come back in the activations' type, gradients of the
    parameters in fp32."""

    @staticmethod
    def forward(ctx, A1, W1, A2, W2, bias, out_f32):
        ctx.save_for_backward(A1, W1, A2, W2)
        ctx.has_bias = bias is not None
        return ops.linear_fwd(A1, W1, A2, W2, bias, out_dtype=torch.float32 if out_f32 else None)

    @staticmethod
    def backward(ctx, g):
        A1, W1, A2, W2 = ctx.saved_tensors
        g = g.contiguous()
        need = ctx.needs_input_grad
        gb = g
        if A1.dtype == torch.bfloat16 and g.dtype == torch.float32:
            gb = ops.cast_to_bf16(g)[:, :g.size(1)]     # gradient of fp32 logits entering the bf16 part of the graph
        dA1 = ops.linear_fwd(gb, W1.t().contiguous()) if need[0] else None
        dW1 = ops.linear_wgrad(g, A1) if need[1] else None
        dA2 = ops.linear_fwd(gb, W2.t().contiguous()) if (A2 is not None and need[2]) else None
        dW2 = ops.linear_wgrad(g, A2) if (A2 is not None and need[3]) else None
        db = ops.colsum(g) if (ctx.has_bias and need[4]) else None
        return dA1, dW1, dA2, dW2, db, None


class _BatchNormAct(torch.autograd.Function):
    """y = act(BatchNorm1d(x)); train mode uses batch statistics and updates the running buffers."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        if training:
            mean, var = ops.bn_batch_stats(x, running_mean, running_var, momentum)
        else:
            mean, var = running_mean, running_var
        scale, shift = ops.bn_fold(gamma, beta, mean, var, eps)
        y = ops.scale_shift_act(x, scale, shift, relu)
        ctx.save_for_backward(x, y, gamma, mean, var)
        ctx.cfg = (bool(training), float(eps), bool(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, var = ctx.saved_tensors
        training, eps, relu = ctx.cfg
        dx, dgamma, dbeta = ops.bn_relu_bwd(x, y, dy.contiguous(), gamma, mean, var, eps, training, relu)
        return dx, dgamma, dbeta, None, None, None, None, None, None


class _SageTrainLayer(torch.autograd.Function):
    """y = relu(BatchNorm_train(lin_j(mean_j x_j * lin_e(e_ji)) + lin_i(x[:n_dst])))  -- conv + norm + ReLU of one training-mode
    layer (reference :214-219) as one library call forward and one backward (csrc/train.hip issues the same kernels as the
    Functions above, in the same order: bit-identical results, a fraction of the host time).  plan None: the decoder's
    Linear + BatchNorm + ReLU block (:180-186)."""

    @staticmethod
    def forward(ctx, x, edge_attr, We, be, Wj, bj, Wi, gamma, beta, bn, plan, relu, scene_rows):
        n_dst = plan.n_dst if plan is not None else x.size(0)
        # scene_rows: `edge_attr` is the SCENE's tensor and plan.edge_rows (the block's e_id) selects its rows inside the kernels
        # -- the values of edge_attr[e_id] without the per-layer copy
        parts = (plan.rowptr, plan.src, plan.edge_rows if scene_rows else plan.eid) if plan is not None else None
        y, a, z, stats = ops.sage_layer_train_fwd(parts, n_dst, x, edge_attr, We, be, Wj, bj, Wi, gamma, beta, bn.running_mean, bn.running_var,
                                                  bn.momentum, bn.eps, relu)   # numeric momentum: sage_train_layer_supported
        ctx.plan = plan
        ctx.edge_index = plan.edge_index if plan is not None else None   # the lazily built transposed plan (backward) reads it
        ctx.cfg = (float(bn.eps), bool(relu), bj is not None, bool(scene_rows))
        ctx.save_for_backward(x, edge_attr, We, be, Wj, Wi, gamma, a, z, y, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, edge_attr, We, be, Wj, Wi, gamma, a, z, y, stats = ctx.saved_tensors
        eps, relu, has_bias, scene_rows = ctx.cfg
        plan = ctx.plan
        t_parts = rowptr = None
        n_src = n_dst = x.size(0)
        if plan is not None:
            t_parts, rowptr, n_src, n_dst = plan.transposed, plan.rowptr, plan.n_src, plan.n_dst
            if scene_rows:
                t_parts = (t_parts[0], t_parts[1], plan.transposed_edge_rows)
        dx, dWe, dbe, dWj, dbj, dWi, dgamma, dbeta = ops.sage_layer_train_bwd(
            t_parts, rowptr, n_src, n_dst, x, edge_attr, We, be, Wj, Wi, has_bias, gamma, stats, eps, relu, a, z, y, dy.contiguous(),
            ctx.needs_input_grad[0])
        return dx, None, dWe, dbe, dWj, dbj, dWi, dgamma, dbeta, None, None, None, None


def sage_train_layer(x, plan, edge_attr, lin_e, lin_j, lin_i, bn: torch.nn.BatchNorm1d, relu: bool = True, scene_rows: bool = False):
    """One training-mode layer through the composite entry points; callers check `sage_train_layer_supported` first.
    `scene_rows`: edge_attr is the scene's [E_all, F] tensor, read through plan.edge_rows (see GraphPlan.edge_rows)."""
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    We, be = (lin_e.weight, lin_e.bias) if lin_e is not None else (None, None)
    return _SageTrainLayer.apply(x, edge_attr if lin_e is not None else None, We, be, lin_j.weight, lin_j.bias,
                                 lin_i.weight if lin_i is not None else None, bn.weight, bn.bias, bn, plan, relu, scene_rows)


def sage_train_layer_supported(x, lin_e, bn) -> bool:
    """fp32 or bf16 activations with unit column stride, lin_e a single Linear over <= 32 attributes (or None), BatchNorm in training mode
    with affine parameters and running buffers; everything else runs through the separate Functions."""
    return (ops.TRAIN_COMPOSITE and x.dtype in ops.ACT and x.dim() == 2 and x.stride(1) == 1 and x.size(0) > 0
            and (lin_e is None or (isinstance(lin_e, torch.nn.Linear) and lin_e.in_features <= 32 and lin_e.bias is not None))
            and isinstance(bn, torch.nn.BatchNorm1d) and bn.training and bn.affine and bn.running_mean is not None
            and bn.momentum is not None)   # momentum=None (cumulative average, factor 1/num_batches_tracked) runs through batch_norm_act


class _StaticTrainModel(torch.autograd.Function):
    """All conv layers of the Static model (+ the decoder's Linear + BatchNorm + ReLU block) in training mode: one library call
    forward, one backward (dgnn_static_train_fwd / _bwd = the per-layer composite calls issued back to back from C++).  Tensor
    inputs: x0, then 7 parameters per layer (We, be, Wj, bj, Wi, gamma, beta; None where a layer has none)."""

    @staticmethod
    def forward(ctx, x0, spec, *params):
        layers = []
        for i, sp in enumerate(spec):
            We, be, Wj, bj, Wi, gamma, beta = params[7 * i:7 * i + 7]
            plan = sp["plan"]
            # addresses only (the library calls take pointers and counts): no views of the block builder's buffers are cut for them
            layers.append(dict(plan_parts=plan.part_ptrs(bool(sp["scene_rows"])) if plan is not None else None,
                               n_dst=plan.n_dst if plan is not None else sp["n_rows"], n_src=plan.n_src if plan is not None else sp["n_rows"],
                               edge_attr=sp["edge_attr"] if We is not None else None, We=We, be=be, Wj=Wj, bj=bj, Wi=Wi, gamma=gamma, beta=beta, bn=sp["bn"]))
        y, buf, meta = ops.static_train_fwd(x0, layers)
        ctx.layers, ctx.meta, ctx.spec = layers, meta, spec
        ctx.edge_indices = [sp["plan"].edge_index for sp in spec if sp["plan"] is not None]   # lazily built transposed plans read them
        ctx.save_for_backward(x0, buf)
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, buf = ctx.saved_tensors
        for l, sp in zip(ctx.layers, ctx.spec):
            plan = sp["plan"]
            if plan is not None:
                l["t_parts"] = plan.transposed_ptrs(bool(sp["scene_rows"]))
        grads = ops.static_train_bwd(x0, ctx.layers, buf, ctx.meta, dy.contiguous())
        out = [None, None]
        for g in grads:
            out += list(g)
        return tuple(out)


def static_train_model(x0, spec):
    """spec: per layer dict(plan | None, n_rows (plain block), edge_attr, scene_rows, lin_e | None, lin_j, lin_i | None, bn | None)"""
    params = []
    for sp in spec:
        le, lj, li, bn = sp["lin_e"], sp["lin_j"], sp["lin_i"], sp["bn"]
        params += [le.weight if le is not None else None, le.bias if le is not None else None, lj.weight, lj.bias, li.weight if li is not None else None,
                   bn.weight if bn is not None else None, bn.bias if bn is not None else None]   # bn None: a plain Linear (decoder output)
    return _StaticTrainModel.apply(x0, spec, *params)


class _SageUpdatedLayer(torch.autograd.Function):
    """(y, phi) of one Updated-variant conv (reference surfaceNetUpdatedEdgeFilters.py:147-170) with the ReLU that follows it
    (:239-247), one library call forward and one backward (csrc/train.hip); fp32 or bf16 storage."""

    @staticmethod
    def forward(ctx, x, ea, We, be, Wl, bl, Wr, plan, relu):
        y, phi, a = ops.sage_updated_train_fwd((plan.rowptr, plan.src, plan.eid), plan.n_dst, x, ea, We, be, Wl, bl, Wr, relu)
        ctx.plan = plan
        ctx.edge_index = plan.edge_index   # the lazily built transposed plan (backward) reads it
        ctx.cfg = (bool(relu), bl is not None)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, ea, We, Wl, Wr, phi, a, y)
        return y, phi

    @staticmethod
    def backward(ctx, dy, dphi_ext):
        x, ea, We, Wl, Wr, phi, a, y = ctx.saved_tensors
        relu, has_bias = ctx.cfg
        plan = ctx.plan
        if dy is None:
            dy = torch.zeros_like(y)
        dx, d_ea, dWe, dbe, dWl, dbl, dWr = ops.sage_updated_train_bwd(
            plan.transposed, plan.rowptr, plan.n_src, plan.n_dst, x, ea, We, Wl, Wr, has_bias, relu, phi, a, y, dy.contiguous(),
            dphi_ext.contiguous() if dphi_ext is not None else None, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, d_ea, dWe, dbe, dWl, dbl, dWr, None, None


def sage_updated_layer(x, plan, ea, lin_e, lin_l, lin_r, relu):
    return _SageUpdatedLayer.apply(x, ea, lin_e.weight, lin_e.bias, lin_l.weight, lin_l.bias, lin_r.weight if lin_r is not None else None, plan, relu)


class _UpdatedConvStack(torch.autograd.Function):
    """All conv layers of the Updated variant with the edge chaining between them (reference surfaceNetUpdatedEdgeFilters.py:229-243): one library
    call forward, one backward (dgnn_updated_stack_fwd / _bwd = the per-layer composite calls and the chaining issued back to back from C++).
    Tensor inputs: x0, then (We, be, Wl, bl, Wr) per layer (None where a layer has none)."""

    @staticmethod
    def forward(ctx, x0, edge_attr_all, pos, spec, tail, *params):
        layers = []
        for i, sp in enumerate(spec):
            We, be, Wl, bl, Wr = params[5 * i:5 * i + 5]
            layers.append(dict(plan=sp["plan"], e_id=sp["e_id"], rows0=sp.get("rows0"), edge_in=sp["edge_in"], relu=sp["relu"], We=We, be=be, Wl=Wl, bl=bl, Wr=Wr))
        y, saved = ops.updated_stack_fwd(x0, edge_attr_all, pos, layers)
        ctx.layers, ctx.saved = layers, saved
        ctx.edge_indices = [sp["plan"].edge_index for sp in spec]     # lazily built transposed plans read them
        ctx.tail = None
        if tail:        # out_net (Linear, ReLU, Linear -> fp32 logits) inside the same node
            W1, b1, W3, b3 = params[5 * len(spec):5 * len(spec) + 4]
            logits, h = ops.updated_tail_fwd(y, W1, b1, W3, b3)
            ctx.tail = (y, h, W1, W3)
            ctx.save_for_backward(x0)
            return logits
        ctx.save_for_backward(x0)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x0,) = ctx.saved_tensors
        tail_grads = []
        dy = dy.contiguous()
        if ctx.tail is not None:
            y, h, W1, W3 = ctx.tail
            dy, dW1, db1, dW3, db3 = ops.updated_tail_bwd(y, W1, W3, h, dy if dy.dtype == torch.float32 else dy.float())
            tail_grads = [dW1, db1, dW3, db3]
        grads = ops.updated_stack_bwd(x0, ctx.layers, ctx.saved, dy)
        out = [None, None, None, None, None]
        for g in grads:
            out += list(g)
        return tuple(out + tail_grads)


def updated_conv_stack(x0, edge_attr_all, pos, spec, out_net=None):
    """spec: per layer dict(plan, e_id int64 [E_l], rows0 (layer 0: e_id as int32), edge_in, relu, lin_e, lin_l, lin_r | None).
    `out_net` = (Linear, Linear) with biases: the model's output network behind the stack (fp32 logits come back)."""
    params = []
    for sp in spec:
        le, ll, lr = sp["lin_e"], sp["lin_l"], sp["lin_r"]
        params += [le.weight, le.bias, ll.weight, ll.bias, lr.weight if lr is not None else None]
    if out_net is not None:
        params += [out_net[0].weight, out_net[0].bias, out_net[1].weight, out_net[1].bias]
    return _UpdatedConvStack.apply(x0, edge_attr_all, pos, spec, out_net is not None, *params)


def sage_updated_layer_supported(x, ea, lin_e) -> bool:
    return (ops.TRAIN_COMPOSITE and x.dim() == 2 and x.dtype in ops.ACT and x.stride(1) == 1 and x.size(0) > 0 and ea.dim() == 2 and ea.dtype == x.dtype
            and ea.stride(1) == 1 and lin_e.bias is not None and (x.dtype == torch.float32 or (x.size(1) % 2 == 0 and x.stride(0) % 2 == 0)))


class _KLCellLoss(torch.autograd.Function):
    """volume-weighted KL cell loss (reference learning/runModel.py:171-209) -> (loss, sums[3] = sum cell*w, sum w, OA count)"""

    @staticmethod
    def forward(ctx, logits, gt, vol, norm):
        loss, sums = ops.kl_cell_loss_fwd(logits, gt, vol, norm)
        ctx.save_for_backward(logits, gt, vol, sums)
        ctx.norm = norm
        ctx.mark_non_differentiable(sums)
        return loss, sums

    @staticmethod
    def backward(ctx, g, _):
        logits, gt, vol, sums = ctx.saved_tensors
        return ops.kl_cell_loss_bwd(logits, gt, vol, ctx.norm, sums, g), None, None, None


def kl_cell_loss(logits, gt, vol, cell_norm=None):
    return _KLCellLoss.apply(logits, gt, vol, ops.CELL_NORMS.get(cell_norm, 0))


class _SceneBatchNormRelu(torch.autograd.Function):
    """BatchNorm1d (training mode) + ReLU over a scene that is cut across ranks (dgnn_amd/partition.py; SURVEY 8e: one [2 C] all-reduce per layer each
    way): per-row work in the library's kernels -- local column statistics (dgnn_bn_batch_stats), scale / shift + ReLU (dgnn_scale_shift_act), the
    backward's column sums and apply (dgnn_bn_relu_bwd_sums / _apply) --, the [C]-vectors in between on the host side of the collective in fp64."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, reduce, relu):
        n_loc, c = x.shape
        if n_loc == 0:
            # a rank that owns no rows of this scene (rcb_partition of a tiny scene) contributes zero sums and STILL enters both
            # all-reduces -- the row kernels require M > 0 and raising before the collective left the other ranks waiting in it
            st = torch.zeros(2 * c + 1, dtype=torch.float64, device=x.device)
        else:
            m_l, v_l = ops.bn_batch_stats(x)                              # local mean / biased variance
            m64 = m_l.double()
            st = torch.cat([m64 * n_loc, (v_l.double() + m64 * m64) * n_loc, torch.full((1,), float(n_loc), dtype=torch.float64, device=x.device)])
        st = reduce(st)
        n = float(st[2 * c].item())
        mean64 = st[:c] / n
        var64 = (st[c:2 * c] / n - mean64 * mean64).clamp_min(0.0)
        mean, var = mean64.float(), var64.float()
        if bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                bn.num_batches_tracked.add_(1)
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - mom).add_((mom * mean64).to(bn.running_mean.dtype))
                bn.running_var.mul_(1 - mom).add_((mom * var64 * (n / max(n - 1.0, 1.0))).to(bn.running_var.dtype))
        if n_loc == 0:
            y = torch.empty_like(x)
        else:
            scale, shift = ops.bn_fold(gamma, beta, mean, var, bn.eps)
            y = ops.scale_shift_act(x, scale, shift, relu)
        ctx.save_for_backward(x, y, gamma, mean, var)
        ctx.cfg = (reduce, bool(relu), float(bn.eps), n)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, var = ctx.saved_tensors
        reduce, relu, eps, n = ctx.cfg
        dy = dy.contiguous()
        empty = x.size(0) == 0                                              # (see forward: zero sums, both collectives entered, nothing to apply)
        loc = torch.zeros((2, x.size(1)), dtype=torch.float32, device=x.device) if empty else ops.bn_relu_bwd_sums(x, y, dy, mean, var, eps, relu)   # this rank's (sum g, sum g * x_hat) = its dbeta / dgamma terms
        glob = reduce(loc.double().reshape(-1).clone()).float().reshape(2, -1)
        dx = torch.empty_like(x) if empty else ops.bn_relu_bwd_apply(x, y, dy, gamma, mean, var, eps, relu, glob, n)
        return dx, loc[1].clone(), loc[0].clone(), None, None, None


def scene_batch_norm_relu(x, bn: torch.nn.BatchNorm1d, reduce, relu: bool = True):
    """`reduce(t)`: all-reduce (sum) of a 1-D fp64 tensor over the ranks that share the scene"""
    return _SceneBatchNormRelu.apply(x, bn.weight, bn.bias, bn, reduce, relu)


def aggregate(x_src, plan, edge_attr=None, We=None, be=None, phi=None):
    if We is not None:
        return _Aggregate.apply(x_src, edge_attr, We, be, plan)
    if phi is not None:
        return _AggregatePhi.apply(x_src, phi, plan)
    return _AggregatePlain.apply(x_src, plan)


def linear2(A1, W1, A2=None, W2=None, bias=None, out_f32=False):
    return _Linear2.apply(A1, W1, A2, W2, bias, out_f32)


class _ToBF16(torch.autograd.Function):
    """fp32 -> bf16 storage (inputs entering the bf16 part of the graph); the gradient is widened back"""

    @staticmethod
    def forward(ctx, x):
        return ops.cast_to_bf16(x, x.size(1) if x.size(1) % 2 == 0 else None)[:, :x.size(1)]

    @staticmethod
    def backward(ctx, g):
        return ops.cast_to_f32(g.contiguous())


def to_bf16(x):
    return x if x.dtype == torch.bfloat16 else _ToBF16.apply(x)


def batch_norm_act(x, bn: torch.nn.BatchNorm1d, relu: bool):
    momentum = bn.momentum if bn.momentum is not None else 0.0
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:   # torch.nn.BatchNorm1d: cumulative moving average, factor 1 / num_batches_tracked (after this batch)
            momentum = 1.0 / float(bn.num_batches_tracked)
    return _BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, momentum, bn.eps, relu)


class _ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.relu(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ops.relu_bwd(y, g)


def relu(x):
    return _ReLU.apply(x)


class _GatherRows(torch.autograd.Function):
    """out = src[idx, :cols]  (rows; idx unique)"""

    @staticmethod
    def forward(ctx, src, idx, cols):
        ctx.save_for_backward(idx)
        ctx.shape = (src.size(0), src.size(1))
        return ops.gather_rows(src, idx.to(torch.int32), cols)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        ops.scatter_rows_(out[:, :g.size(1)], idx, g.contiguous())
        return out, None, None


class _ChainEdges(torch.autograd.Function):
    """next layer's edge rows out of this layer's phi: relu(zeros[E_all, C]; [e_id_cur] = phi)[e_id_next, :c] (reference
    surfaceNetUpdatedEdgeFilters.py:233-241) without the [E_all, C] tensor; see csrc/chain.hip"""

    @staticmethod
    def forward(ctx, phi, e_id_cur, e_id_next, c, pos, relu):
        out, inv = ops.edge_chain_fwd(phi, e_id_cur, e_id_next, c, pos, relu)
        ctx.save_for_backward(phi, inv)
        ctx.cfg = (int(c), bool(relu))
        return out

    @staticmethod
    def backward(ctx, g):
        phi, inv = ctx.saved_tensors
        c, relu = ctx.cfg
        return ops.edge_chain_bwd(g.contiguous(), phi, inv, c, relu), None, None, None, None, None


def chain_edges(phi, e_id_cur, e_id_next, c, pos, relu=True):
    return _ChainEdges.apply(phi, e_id_cur, e_id_next, c, pos, relu)


class _ScatterRows(torch.autograd.Function):
    """out = zeros[n_rows, C]; out[idx] = src  (idx unique)"""

    @staticmethod
    def forward(ctx, src, idx, n_rows):
        ctx.save_for_backward(idx)
        out = torch.zeros((n_rows, src.size(1)), dtype=src.dtype, device=src.device)
        return ops.scatter_rows_(out, idx, src)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return ops.gather_rows(g.contiguous(), idx.to(torch.int32)), None, None


def gather_rows(src, idx, cols=None):
    return _GatherRows.apply(src, idx, cols)


def scatter_rows(src, idx, n_rows):
    return _ScatterRows.apply(src, idx, n_rows)
