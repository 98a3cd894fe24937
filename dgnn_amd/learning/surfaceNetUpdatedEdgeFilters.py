"""MI355X drop-in for the coherent part of the reference's learning/surfaceNetUpdatedEdgeFilters.py.

``SAGEConv(in_channels, out_channels, edge_in_channels)`` returns ``(out, phi)`` (reference :147-170)
and ``SurfaceNet(n_node_features, clf).forward(data_all)`` chains the edge embeddings from layer to
layer (:216-251).  State-dict keys: ``convs.N.lin_l.{weight,bias}``, ``convs.N.lin_r.weight``,
``convs.N.lin_e.{weight,bias}``, ``out_net.{1,3}.{weight,bias}``.

phi feeds both this layer's aggregation and the next layer's lin_e, so it is materialised
([E_l, C_in], written once, read twice) and differentiated through both uses; the dense lin_e runs on
the fp32 MFMA GEMM (its K is 2, F or h_{k-2}) and the aggregation kernel takes phi rows as given.

The reference's three ``inference_*`` methods of this file call the conv without ``edge_attr``
(:281,:317,:352) and raise there; they are not part of the path and raise here too.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import Linear

from .. import functional as Fn
from ..graph import GraphPlan, plan_for


# materialise the whole-scene [E_all, C] edge tensor per layer as the reference does (DGNN_CHAIN_DENSE=1) instead of chaining
# only the rows the next layer reads; same values, kept for A/B tests
CHAIN_SPARSE = __import__("os").environ.get("DGNN_CHAIN_DENSE", "0") != "1"


def _dev_f32(t, device):
    if t.device != torch.device(device) or t.dtype != torch.float32:
        t = t.to(device=device, dtype=torch.float32)
    return t


class SAGEConv(nn.Module):

    def __init__(self, in_channels, out_channels, edge_in_channels, normalize=False, bias=True, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.edge_in_channels = edge_in_channels
        self.out_channels = out_channels
        self.normalize = normalize
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.lin_l = Linear(in_channels[0], out_channels, bias=bias)
        self.lin_r = Linear(in_channels[1], out_channels, bias=False)
        self.lin_e = Linear(edge_in_channels, in_channels[0], bias=bias)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()
        self.lin_e.reset_parameters()

    def forward(self, x, edge_attr, edge_index, size=None, plan: GraphPlan = None):
        if isinstance(x, torch.Tensor):
            x = (x, x)
        x_src, x_dst = x
        if plan is None:
            plan = plan_for(edge_index, x_src.size(0), x_dst.size(0))
        if x_src.dtype == torch.bfloat16:
            edge_attr = Fn.to_bf16(edge_attr)                                        # bf16 storage: phi is written once in bf16
        phi = Fn.linear2(edge_attr, self.lin_e.weight, bias=self.lin_e.bias)        # :156
        a = Fn.aggregate(x_src, plan, phi=phi)                                       # :158 (mean of x_j * phi)
        out = Fn.linear2(a, self.lin_l.weight, x_dst, self.lin_r.weight, self.lin_l.bias)  # :159-165
        if self.normalize:
            out = torch.nn.functional.normalize(out, p=2., dim=-1)                   # :167-168 (never enabled by a config)
        return out, phi

    def __repr__(self):
        return '{}(in:{}, edge_in:{}, edge_out:{}, out:{})'.format(self.__class__.__name__, self.in_channels,
                                                                   self.edge_in_channels, self.in_channels, self.out_channels)


class SurfaceNet(nn.Module):

    def __init__(self, n_node_features, clf):
        super().__init__()
        self.clf = clf
        self.n_classes = 2
        self.n_node_feat = n_node_features
        p = clf.training.model_params
        self.convs = nn.ModuleList()
        self.convs.append(SAGEConv(self.n_node_feat, p[0], 2))
        self.convs.append(SAGEConv(p[0], p[1], self.n_node_feat))
        for i in range(len(p) - 2):
            self.convs.append(SAGEConv(p[i + 1], p[i + 2], p[i], normalize=False))
        self.num_layers = len(self.convs)
        if clf.training.model_name[-1] == "+":
            self.out_net = nn.Sequential(nn.ReLU(True), nn.Linear(p[-1], 128), nn.ReLU(True), nn.Linear(128, 2))

    storage_dtype = torch.float32
    _chain_pos = None

    def _chain_table(self, n_edges, dev):
        """[E_all] int32 position table of the edge chaining, all -1 between uses (one per model and device, 4 bytes per scene edge)"""
        t = self._chain_pos
        if t is None or t.numel() != n_edges or t.device != torch.device(dev):
            t = self._chain_pos = torch.full((n_edges,), -1, dtype=torch.int32, device=dev)
        return t

    def set_storage_dtype(self, dtype):
        """torch.float32 (default) or torch.bfloat16: node activations and the chained edge embeddings phi are stored in bf16
        (phi is the [E, C] tensor that is written once and read twice per layer -- where bf16 halves the traffic), products on
        the bf16 matrix cores with fp32 accumulation, parameters and their gradients fp32 (BASELINE config 3)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("storage dtype must be torch.float32 or torch.bfloat16")
        self.storage_dtype = dtype
        return self

    def forward(self, data_all):
        dev = self.clf.temp.device
        if not str(dev).startswith("cuda"):
            raise RuntimeError("clf.temp.device=%r: dgnn_amd runs on a GPU only (no CPU fallback)" % (dev,))
        f = self.clf.features
        x_all = data_all.x
        n_id = data_all.n_id.to(x_all.device)
        from ..sampler import block_rows
        col0 = 1 if (f.normalization_feature and not f.keep_normalization_feature) else 0
        x = block_rows(n_id, x_all, col0, x_all.size(1) - col0, "all") if x_all.dim() == 2 else None     # gathered by the block's builder (attach_rows)?
        if x is None:
            x = x_all[n_id, 1:] if col0 else x_all[n_id, :]
        x = _dev_f32(x, dev)
        if self.storage_dtype == torch.bfloat16:
            x = Fn.to_bf16(x)
        edge_attr = _dev_f32(data_all.edge_attr, dev)
        n_edges = edge_attr.size(0)
        phi = e_prev = None
        plus = self.clf.training.model_name[-1] == "+"
        self._stack_did_tail = False
        x_stack = self._conv_stack(x, edge_attr, data_all.adjs, dev, plus)
        for i in range(self.num_layers if x_stack is None else 0):
            edge_index, e_id, size = data_all.adjs[i]
            e_id = e_id.to(dev)
            conv = self.convs[i]
            if i == 0:
                ea = Fn.gather_rows(edge_attr, e_id, conv.edge_in_channels)           # :237 edge_attr[e_id, :edge_in]
            elif CHAIN_SPARSE:
                # :236-241 zeros[E_all, C]; [e_prev] = phi; relu; [e_id, :edge_in] -- computed for the rows that are read
                ea = Fn.chain_edges(phi, e_prev, e_id, conv.edge_in_channels, self._chain_table(n_edges, dev), relu=True)
            else:
                ea = Fn.gather_rows(Fn.relu(Fn.scatter_rows(phi, e_prev, n_edges)), e_id, conv.edge_in_channels)
            last = i == self.num_layers - 1
            relu_after = (not last) or plus        # :239-241 between layers; :245-246 F.relu + out_net[0] after the last one of "sage+"
            ea_in = Fn.to_bf16(ea) if x.dtype == torch.bfloat16 else ea               # bf16 storage: phi is written once in bf16
            if not conv.normalize and Fn.sage_updated_layer_supported(x, ea_in, conv.lin_e):
                # conv (+ the ReLU that follows it) as one library call forward, one backward
                edge_index = edge_index.to(dev)
                x, phi = Fn.sage_updated_layer(x, plan_for(edge_index, x.size(0), size[1]), ea_in, conv.lin_e, conv.lin_l, conv.lin_r, relu_after)
            else:
                x, phi = conv((x, x[:size[1]]), ea, edge_index.to(dev))
                if relu_after:
                    x = Fn.relu(x)  # after the last layer: F.relu(x) and out_net[0], another ReLU -- relu(relu(x)) == relu(x)
            e_prev = e_id
        if x_stack is not None:
            x = x_stack
            if self._stack_did_tail:
                return x
        # (the last layer's new_edge_attr, :236-237, is never read)
        if plus:                                                                      # :245-247
            x = Fn.linear2(x, self.out_net[1].weight, bias=self.out_net[1].bias)
            x = Fn.relu(x)
            x = Fn.linear2(x, self.out_net[3].weight, bias=self.out_net[3].bias, out_f32=True)
        return x.float() if x.dtype == torch.bfloat16 else x

    def _conv_stack(self, x, edge_attr, adjs, dev, plus, spec_only=False):
        """All conv layers and the edge chaining between them through ONE library call each way (Fn.updated_conv_stack) when every layer takes the
        composite form: sparse chaining, no output normalisation, lin_e with a bias, even widths in bf16 storage.  Returns the last layer's
        activations, or None (the per-layer path then runs)."""
        from .. import ops
        if not (ops.UPDATED_STACK and ops.TRAIN_COMPOSITE and CHAIN_SPARSE) or self.num_layers > 8 or x.dim() != 2 or x.size(0) == 0 or x.stride(1) != 1:
            return None
        bf = x.dtype == torch.bfloat16
        if x.dtype not in ops.ACT or edge_attr.dtype != torch.float32 or edge_attr.dim() != 2 or edge_attr.stride(1) != 1 or not edge_attr.is_cuda:
            return None
        spec, c, n_src = [], x.size(1), x.size(0)
        for i in range(self.num_layers):
            edge_index, e_id, size = adjs[i]
            conv = self.convs[i]
            k = conv.edge_in_channels
            if conv.normalize or conv.lin_e.bias is None or e_id is None or size[0] != n_src or conv.lin_l.in_features != c or (bf and (c % 2 or k % 2)) \
                    or (i == 0 and k > edge_attr.size(1)) or (i > 0 and k > self.convs[i - 1].lin_l.in_features):
                return None
            if bf and i == 0 and x.stride(0) % 2:
                return None
            e_id = e_id.to(dev)
            plan = plan_for(edge_index.to(dev), size[0], size[1])
            sp = dict(plan=plan, e_id=e_id, edge_in=k, relu=(i < self.num_layers - 1) or plus, lin_e=conv.lin_e, lin_l=conv.lin_l, lin_r=conv.lin_r)
            if i == 0:   # the block builder's plans carry e_id as int32 already
                sp["rows0"] = plan.edge_rows if plan.has_edge_rows else e_id.to(torch.int32)
            spec.append(sp)
            c, n_src = conv.lin_l.out_features, size[1]
        out_net = None
        if plus and ops.UPDATED_TAIL_IN_CALL and isinstance(self.out_net[1], nn.Linear) and isinstance(self.out_net[3], nn.Linear) \
                and self.out_net[1].bias is not None and self.out_net[3].bias is not None and self.out_net[1].in_features == c:
            out_net = (self.out_net[1], self.out_net[3])       # :245-247 inside the same library-call pair; fp32 logits come back
            self._stack_did_tail = True
        if spec_only:
            return spec, out_net
        return Fn.updated_conv_stack(x, edge_attr, self._chain_table(edge_attr.size(0), dev), spec, out_net)

    @torch.no_grad()
    def train_step_direct(self, data_all, loss_fn):
        """Forward, loss and backward of ONE training step without the autograd engine (round 6; reference learning/runModel.py:266-279 on this model's
        forward :216-251): the conv stack's and the output network's library calls each way are issued directly around
        `loss_fn(logits) -> (loss, dlogits)`, and every parameter's .grad is SET to its fresh gradient (what zero_grad + backward leave).  The same
        kernels in the same order as the autograd node (functional._UpdatedConvStack): the same numbers (tests/test_gpu_train.py).  The engine --
        its worker-thread hand-over, the node bookkeeping, three Function.apply calls -- was 0.15 ms of a 0.85 ms host-bound step.
        Returns the detached loss, or None when the model does not take the one-call form with the output network inside (the caller then runs
        forward() / backward())."""
        from .. import ops
        dev = self.clf.temp.device
        if not str(dev).startswith("cuda") or self.clf.training.model_name[-1] != "+":
            return None
        f = self.clf.features
        x_all = data_all.x
        n_id = data_all.n_id.to(x_all.device)
        from ..sampler import block_rows
        col0 = 1 if (f.normalization_feature and not f.keep_normalization_feature) else 0
        x = block_rows(n_id, x_all, col0, x_all.size(1) - col0, "all") if x_all.dim() == 2 else None
        if x is None:
            x = x_all[n_id, 1:] if col0 else x_all[n_id, :]
        x = _dev_f32(x, dev)
        if self.storage_dtype == torch.bfloat16 and x.dtype != torch.bfloat16:
            x = ops.cast_to_bf16(x, x.size(1) if x.size(1) % 2 == 0 else None)[:, :x.size(1)]      # (= Fn.to_bf16's forward)
        edge_attr = _dev_f32(data_all.edge_attr, dev)
        self._stack_did_tail = False
        built = self._conv_stack(x, edge_attr, data_all.adjs, dev, True, spec_only=True)
        if built is None or built[1] is None:
            return None
        spec, (lin1, lin3) = built
        layers = []
        for sp in spec:
            le, ll, lr = sp["lin_e"], sp["lin_l"], sp["lin_r"]
            layers.append(dict(plan=sp["plan"], e_id=sp["e_id"], rows0=sp.get("rows0"), edge_in=sp["edge_in"], relu=sp["relu"], We=le.weight, be=le.bias,
                               Wl=ll.weight, bl=ll.bias, Wr=lr.weight if lr is not None else None))
        y, saved = ops.updated_stack_fwd(x, edge_attr, self._chain_table(edge_attr.size(0), dev), layers)
        logits, h = ops.updated_tail_fwd(y, lin1.weight, lin1.bias, lin3.weight, lin3.bias)
        loss, dlogits = loss_fn(logits)
        dy, dW1, db1, dW3, db3 = ops.updated_tail_bwd(y, lin1.weight, lin3.weight, h, dlogits if dlogits.dtype == torch.float32 else dlogits.float())
        grads = ops.updated_stack_bwd(x, layers, saved, dy)
        written = set()

        def put(p_, g_):
            if p_ is not None and g_ is not None and p_.requires_grad:
                p_.grad = g_
                written.add(id(p_))
        for l, g in zip(layers, grads):
            for name, gr in zip(("We", "be", "Wl", "bl", "Wr"), g):
                put(l[name], gr)
        put(lin1.weight, dW1), put(lin1.bias, db1), put(lin3.weight, dW3), put(lin3.bias, db3)
        plist = self.__dict__.get("_param_list")
        if plist is None:
            plist = self.__dict__["_param_list"] = list(self.parameters())
        for p_ in plist:          # a trainable parameter this step did not reach has NO gradient (what zero_grad + backward leave)
            if p_.requires_grad and p_.grad is not None and id(p_) not in written:
                p_.grad = None
        return loss.detach()

    def _unsupported(self, *a, **k):
        raise NotImplementedError("the reference's surfaceNetUpdatedEdgeFilters.inference_* methods call the conv without "
                                  "edge_attr and cannot run (reference :281,:317,:352); use forward(data_all)")

    inference_batch_layer = inference_layer_batch = inference_layer = _unsupported
