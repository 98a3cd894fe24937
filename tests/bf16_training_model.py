"""Storage-rounding model of the Updated variant's TRAINING step in bf16 storage (test infrastructure; the training twin of
tests/test_bf16_rounding_model_cpu.py -- VERDICT r4 item 1b).

The reference step (learning/surfaceNetUpdatedEdgeFilters.py:147-176, 216-251 forward; autograd backward, learning/runModel.py:279) restated in fp64 on
the CPU with EXPLICIT forward and backward passes, so that a round-to-bf16 can be injected at exactly the tensors the HIP path stores or feeds to the
bf16 matrix cores (dgnn_amd/csrc/train.hip: updated_fwd / updated_bwd, dgnn_updated_stack_fwd / _bwd; bf16 twins of the generic kernels: widen on
load, fp32 arithmetic, round to nearest even on store):

  site "x"    the input rows x[n_id, 1:] cast once (functional._ToBF16) and layer 0's edge rows edge_attr[e_id, :2] (dgnn_cast_f32_to_bf16)
  site "w"    every fp32 master weight as it is staged into a bf16 matrix product (forward: We, Wl, Wr; backward: Wl^T, Wr^T, We^T)
  site "phi"  phi_l = lin_e(ea_l), written once in bf16 (read by the aggregation, by the next layer's lin_e through the chaining and by the backward)
  site "a"    a_l = mean_j x_j * phi, written by the aggregate kernel
  site "y"    y_l = relu?(lin_l(a) + lin_r(x_dst)), GEMM epilogue; the last layer's y are the logits (returned as y.float())
  site "dy"   the gradient arriving at the logits (autograd of y.float(): cast to bf16), every [da | dz.Wr] the input-gradient GEMM stores, every dx the
              aggregate backward stores (dx = agg_bwd(da, phi) + the stored dz.Wr, rounded once)
  site "dphi" dphi_l as the aggregate backward stores it, the sum dphi + dphi_ext (k_add_inplace), d_ea = dphi . We (the gradient the chaining hands to
              the previous layer's phi)

Weight GRADIENTS are fp32 accumulations of exact bf16 x bf16 products and are never rounded.  With every site off the function is the un-rounded
step: tests/test_bf16_rounding_model_cpu.py::test_training_model_without_rounding_is_the_oracle holds it to the oracle's autograd in fp64, which pins
this restatement.  With all sites on it is the yardstick the GPU gradients are compared with (tests/test_gpu_scale.py): whatever separates the HIP
path from THIS model beyond fp32-accumulation noise is a kernel bug; what separates this model from the un-rounded step is the price of the format
(BASELINE.md section 4).
"""
from __future__ import annotations

import torch

ALL_SITES = frozenset(("x", "w", "phi", "a", "y", "dy", "dphi"))


def _rb(t: torch.Tensor) -> torch.Tensor:
    """fp64 -> fp32 -> bf16 (round to nearest even twice: the GPU computes in fp32 and rounds on store) -> fp64"""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def updated_step(sd, params, n_node_feat, x_all, edge_attr_all, n_id, adjs, G, sites=ALL_SITES, drop_col0=True, arith=torch.float64):
    """One forward + backward of the Updated "sage" model (no out_net) on one sampled block.
    sd: state dict (convs.{l}.lin_{e,l,r}.{weight,bias}); adjs[l] = (edge_index [2,E_l] local ids, e_id [E_l] rows of the scene's edge tensors,
    (n_src, n_dst)); loss = (logits * G).sum().  -> (logits fp64 [n_dst_last, params[-1]], {parameter name: gradient fp64})."""
    sites = frozenset(sites)
    R = {s: (_rb if s in sites else (lambda t: t)) for s in ALL_SITES}
    L = len(params)
    f64 = lambda t: t.detach().to("cpu", torch.float64).to(arith)
    P = {k: f64(v) for k, v in sd.items()}
    x = f64(x_all)[n_id.cpu()]
    x = R["x"](x[:, 1:] if drop_col0 else x)
    E_all = edge_attr_all.shape[0]
    saved = []
    ea = R["x"](f64(edge_attr_all)[adjs[0][1].cpu(), :2])
    for l in range(L):
        edge_index, e_id, size = adjs[l]
        src, dst, e_id = edge_index[0].cpu(), edge_index[1].cpu(), e_id.cpu()
        n_dst = int(size[1])
        We, be = P["convs.%d.lin_e.weight" % l], P["convs.%d.lin_e.bias" % l]
        Wl, bl, Wr = P["convs.%d.lin_l.weight" % l], P["convs.%d.lin_l.bias" % l], P["convs.%d.lin_r.weight" % l]
        c_in = Wl.shape[1]
        assert ea.shape[1] == We.shape[1] and x.shape[1] == c_in
        phi = R["phi"](ea @ R["w"](We).t() + be)                                              # :156
        cnt = torch.bincount(dst, minlength=n_dst).clamp(min=1).to(arith)
        S = torch.zeros((n_dst, c_in), dtype=arith).index_add_(0, dst, x[src] * phi)     # :158 (mean of x_j * phi)
        a = R["a"](S / cnt[:, None])
        z = a @ R["w"](Wl).t() + x[:n_dst] @ R["w"](Wr).t() + bl                                # :159-165
        relu = l < L - 1                                                                        # :239-241
        y = R["y"](torch.relu(z) if relu else z)
        p = None
        if l < L - 1:
            # :236-241 zeros[E_all, C]; [e_id] = phi; relu; the rows e_id_next the next layer reads (edge_in_{l+1} == this phi's width)
            pos = torch.full((E_all,), -1, dtype=torch.int64)
            pos[e_id] = torch.arange(e_id.numel())
            p = pos[adjs[l + 1][1].cpu()]
            ea_next = torch.where((p >= 0)[:, None], torch.relu(phi[p.clamp(min=0)]), torch.zeros((), dtype=arith))
        saved.append(dict(src=src, dst=dst, n_dst=n_dst, cnt=cnt, x=x, ea=ea, phi=phi, a=a, y=y, relu=relu, We=We, Wl=Wl, Wr=Wr, p_next=p))
        x = y
        if l < L - 1:
            ea = ea_next
    logits = x
    grads = {}
    g = R["dy"](f64(G))
    dphi_ext = None
    for l in range(L - 1, -1, -1):
        s = saved[l]
        dz = g * (s["y"] > 0) if s["relu"] else g
        grads["convs.%d.lin_l.weight" % l] = dz.t() @ s["a"]
        grads["convs.%d.lin_l.bias" % l] = dz.sum(0)
        grads["convs.%d.lin_r.weight" % l] = dz.t() @ s["x"][:s["n_dst"]]
        da = R["dy"](dz @ R["w"](s["Wl"]))                     # one GEMM against the stacked [Wl^T ; Wr^T]: both halves are stored in the storage type
        dxr = R["dy"](dz @ R["w"](s["Wr"]))
        dm = (da / s["cnt"][:, None])[s["dst"]]
        dphi = R["dphi"](dm * s["x"][s["src"]])
        if l > 0:
            dx = torch.zeros_like(s["x"]).index_add_(0, s["src"], dm * s["phi"])
            dx[:s["n_dst"]] += dxr
            dx = R["dy"](dx)
        if dphi_ext is not None:
            dphi = R["dphi"](dphi + dphi_ext)
        grads["convs.%d.lin_e.weight" % l] = dphi.t() @ s["ea"]
        grads["convs.%d.lin_e.bias" % l] = dphi.sum(0)
        dphi_ext = None
        if l > 0:
            d_ea = R["dphi"](dphi @ R["w"](s["We"]))
            prev = saved[l - 1]
            p = prev["p_next"]
            ok = p >= 0
            dphi_ext = torch.zeros_like(prev["phi"])
            dphi_ext[p[ok]] = d_ea[ok] * (prev["phi"][p[ok]] > 0)
            g = dx
    return logits.double(), {k: v.double() for k, v in grads.items()}


def error_table(ref, got):
    """{name: (max |d| / max |ref|, rms d / rms ref)}"""
    out = {}
    for k, r in ref.items():
        d = got[k].double().cpu() - r
        out[k] = (d.abs().max().item() / max(r.abs().max().item(), 1e-300), d.pow(2).mean().sqrt().item() / max(r.pow(2).mean().sqrt().item(), 1e-300))
    return out
