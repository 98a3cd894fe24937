"""MI355X drop-in for the reference's learning/surfaceNetStaticEdgeFilters.py.

Same public surface: ``SAGEConv(lin_i, lin_j, lin_e)`` callable as
``conv((x_src, x_dst), edge_attr, edge_index)`` (reference :66-87) and ``SurfaceNet(clf)`` with
``forward(data)`` (:196-227), ``inference_layer`` (:323-355), ``inference_batch_layer`` (:232-275),
``inference_layer_batch`` (:279-320), ``.convs / .decoder / .num_layers`` and the checkpoint keys
of data/models/kf96/model_best.ptm (``convs.N.conv.lin_{i,j,e}.*``, ``convs.N.norm.module.*``,
``decoder.{0,3}.*``, ``decoder.1.module.*``).  The nn.Linear / BatchNorm1d sub-modules are parameter
containers only -- their ``forward`` is never called; all arithmetic runs in libdgnn_hip.so.

Ownership and errors follow the reference: inputs are never mutated, logits are returned on the
device, config errors ``print`` and ``sys.exit(1)`` (learning/runModel.py:190-191), everything else
raises.  There is no CPU execution path.
"""
from __future__ import annotations

import sys

import torch
import torch.nn as nn
from torch.nn import Linear

from .. import functional as Fn
from .. import ops
from ..graph import GraphPlan, plan_for


class BatchNorm(nn.Module):
    """Key-compatible stand-in for torch_geometric.nn.norm.BatchNorm (wraps BatchNorm1d as .module)."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True)

    def forward(self, x, relu: bool = False):
        return Fn.batch_norm_act(x, self.module, relu)


def _block_x(x_all, n_id, col0, dev):
    """x_all[n_id, col0:] as fp32 rows on `dev` (reference :206) -- the rows the block's builder already gathered on its own stream when the loader was
    told to (sampler.NeighborSampler.attach_rows), else indexed here"""
    from ..sampler import block_rows
    x = block_rows(n_id, x_all, col0, x_all.size(1) - col0, "all") if x_all.dim() == 2 else None
    if x is None:
        x = x_all[n_id, col0:] if col0 else x_all[n_id, :]
    return _dev_f32(x, dev)


def _dev_f32(t: torch.Tensor, device) -> torch.Tensor:
    """H2D copy as the reference's ``.to(self.clf.temp.device)`` does; keeps views of resident data."""
    if t.device != torch.device(device) or t.dtype != torch.float32:
        t = t.to(device=device, dtype=torch.float32)
    return t if (t.dim() != 2 or t.stride(1) == 1) else t.contiguous()


class SAGEConv(nn.Module):
    """Edge-filtered GraphSAGE conv (reference :20-109): out = lin_j(mean_j x_j * lin_e(e_ji)) + lin_i(x_i)."""

    def __init__(self, lin_i, lin_j, lin_e, **kwargs):
        super().__init__()
        self.lin_i = lin_i
        self.lin_j = lin_j
        self.lin_e = lin_e

    def _filter_args(self, edge_attr, bf16=False):
        """lin_e as (edge_attr, We, be) for the fused kernel, or a materialised phi for edge MLPs (`bf16`: phi is
        materialised in the activations' bf16 storage type)."""
        le = self.lin_e
        if le is None:
            return dict()
        if isinstance(le, Linear):
            if le.in_features in (2, 20):
                return dict(edge_attr=edge_attr, We=le.weight, be=le.bias)
            return dict(phi=Fn.linear2(Fn.to_bf16(edge_attr) if bf16 else edge_attr, le.weight, bias=le.bias))
        if bf16:
            edge_attr = Fn.to_bf16(edge_attr)
        # edge_convs == 2 (:131-136): Linear -> norm -> ReLU -> Linear, materialised per edge
        h = Fn.linear2(edge_attr, le[0].weight, bias=le[0].bias)
        h = le[1](h, relu=True) if le[1] is not None else Fn.relu(h)
        return dict(phi=Fn.linear2(h, le[3].weight, bias=le[3].bias))

    def forward(self, x, edge_attr, edge_index, size=None, plan: GraphPlan = None):
        if isinstance(x, torch.Tensor):
            x = (x, x)
        x_src, x_dst = x
        if plan is None:
            plan = plan_for(edge_index, x_src.size(0), x_dst.size(0))
        a = Fn.aggregate(x_src, plan, **self._filter_args(edge_attr, x_src.dtype == torch.bfloat16))
        if x_dst is not None:
            return Fn.linear2(a, self.lin_j.weight, x_dst, self.lin_i.weight, self.lin_j.bias)
        return Fn.linear2(a, self.lin_j.weight, bias=self.lin_j.bias)

    def __repr__(self):
        return '{}:\nW1: {}\nW2: {}\nΦ: {}'.format(self.__class__.__name__, self.lin_i, self.lin_j, self.lin_e)


class SurfaceNet(nn.Module):

    def normLayer(self, size):
        if self.norm_type == 'b':
            return BatchNorm(size)
        elif self.norm_type == 'l':
            print("normalization 'l' (graph LayerNorm) is not supported by the MI355X path")
            sys.exit(1)
        return None

    def sageLayer(self, input, output):
        li = Linear(input, output, bias=False)
        lj = Linear(input, output, bias=True)
        if self.clf.model.edge_convs == 1:
            le = Linear(self.n_edge_feat, input, bias=True)
        elif self.clf.model.edge_convs == 2:
            le = nn.Sequential()
            le.add_module("0", Linear(self.n_edge_feat, int(self.n_edge_feat * 2)))
            le.add_module("1", self.normLayer(int(self.n_edge_feat * 2)))
            le.add_module("2", nn.ReLU(True))
            le.add_module("3", Linear(int(self.n_edge_feat * 2), input))
        else:
            le = None
        return SAGEConv(li, lj, le)

    def __init__(self, clf):
        super().__init__()
        self.clf = clf
        self.n_classes = 2
        self.n_node_feat = clf.temp.num_node_features
        self.n_edge_feat = clf.temp.num_edge_features
        self.norm_type = clf.model.normalization
        self.output_dim = 2 if clf.training.loss == "kl" else 1

        self.convs = nn.ModuleList()
        widths = [self.n_node_feat] + list(clf.model.convs)
        for cin, cout in zip(widths[:-1], widths[1:]):
            layer = nn.Sequential()
            layer.add_module("conv", self.sageLayer(cin, cout))
            layer.add_module("norm", self.normLayer(cout))
            layer.add_module("relu", nn.ReLU(True))
            self.convs.append(layer)
        self.num_layers = len(self.convs)

        self.decoder = nn.Sequential()
        last = clf.model.convs[-1]
        if clf.model.decoder == 1:
            self.decoder.add_module("0", nn.Linear(last, self.output_dim))
        elif clf.model.decoder == 2:
            self.decoder.add_module("0", nn.Linear(last, int(last / 2)))
            self.decoder.add_module("1", self.normLayer(int(last / 2)))
            self.decoder.add_module("2", nn.ReLU(True))
            self.decoder.add_module("3", nn.Linear(int(last / 2), self.output_dim))

    # ------------------------------------------------------------------------------------------
    storage_dtype = torch.float32

    def set_storage_dtype(self, dtype):
        """torch.float32 (default: the reference's arithmetic) or torch.bfloat16 = the bf16 STORAGE path (BASELINE config 3):
        activations are kept in HBM as bf16, every product runs once on the bf16 matrix cores with fp32 accumulation,
        parameters stay fp32.  Tolerance of that path: |dlogit| <= 5e-2 * max(1, |logit|/8), arg-max agreement >= 99.9 %."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("storage dtype must be torch.float32 or torch.bfloat16")
        self.storage_dtype = dtype
        return self

    def activation_dtype(self, i, input_dtype=torch.float32):
        """dtype of the rows conv layer i leaves in HBM in eval mode: the storage type -- or, in bf16 storage (compensated arithmetic), ops.UROWS
        (torch.int16: UNSIGNED rows, half of bf16's storage rounding in the same bytes, ops.py) while the chain of fused layers that starts on the
        caller's fp32 feature rows is unbroken.  Callers that allocate a layer's output themselves (the partitioned forward) ask here."""
        if self.storage_dtype != torch.bfloat16:
            return self.storage_dtype
        if not (ops.BF16_UNSIGNED_ROWS and ops.BF16_MODE == ops.BF16_COMPENSATED and ops.FUSED_ENABLED) or input_dtype != torch.float32:
            return torch.bfloat16
        for j in range(i + 1):
            conv = self.convs[j][0]
            le = conv.lin_e
            fused = isinstance(le, Linear) and le.in_features == 20 and ops.fused_layer_supported_bf16(conv.lin_j.in_features, conv.lin_j.out_features, 20)
            if not fused or (j == 0 and conv.lin_j.in_features > 32) or not isinstance(self.convs[j][2], nn.ReLU):
                return torch.bfloat16
        return ops.UROWS

    def dominant_kernel_name(self, shape):
        return "k_sage_fused_bf16<%d,%d>" % (32 if shape[0] <= 32 else (64 if shape[0] <= 64 else 128), shape[1])

    def _device(self):
        dev = self.clf.temp.device
        if not str(dev).startswith("cuda"):
            raise RuntimeError("clf.temp.device=%r: dgnn_amd.SurfaceNet runs on a GPU only (no CPU fallback)" % (dev,))
        return dev

    def _input_rows(self, x):
        """drop column 0 (the loss-weight column) when regularization.cell_type is set, reference :329-332"""
        return x[:, 1:] if self.clf.regularization.cell_type else x

    def _storage_input(self, x):
        """fp32 input rows as the first conv layer takes them: unchanged in fp32 storage; in bf16 storage the fused first layer
        reads the fp32 features in place (they are never rounded), any other first layer gets a bf16 copy."""
        if self.storage_dtype != torch.bfloat16 or x.dtype == torch.bfloat16:
            return x
        c0 = self.convs[0][0]
        if isinstance(c0.lin_e, Linear) and c0.lin_e.in_features == 20 and \
                ops.fused_layer_supported_bf16(c0.lin_j.in_features, c0.lin_j.out_features, 20, x):
            return x
        return ops.cast_to_bf16(x)

    def _norm_act(self, layer, x):
        """convs[i][1] then convs[i][2] (reference :218-219): BatchNorm (if any) + ReLU, one kernel chain."""
        norm = layer[1] if len(layer) > 1 and isinstance(layer[1], BatchNorm) else None
        if norm is not None:
            return norm(x, relu=True)
        return Fn.relu(x)

    def _decode(self, x):
        dec = self.decoder
        if len(dec) == 0:
            return x
        if len(dec) == 1:
            return Fn.linear2(x, dec[0].weight, bias=dec[0].bias, out_f32=True)
        if isinstance(dec[1], BatchNorm) and Fn.sage_train_layer_supported(x, None, dec[1].module):
            h = Fn.sage_train_layer(x, None, None, None, dec[0], None, dec[1].module)
        else:
            h = Fn.linear2(x, dec[0].weight, bias=dec[0].bias)
            h = dec[1](h, relu=True) if dec[1] is not None else Fn.relu(h)
        return Fn.linear2(h, dec[3].weight, bias=dec[3].bias, out_f32=True)   # logits stay fp32 in the bf16 storage path too

    # ---- TRAIN FORWARD (reference :196-227) ---------------------------------------------------
    def forward(self, data):
        dev = self._device()
        x_all = data.all.x
        n_id = data.batch_n_id.to(x_all.device)
        x = _block_x(x_all, n_id, 1 if self.clf.regularization.cell_type else 0, dev)
        if self.storage_dtype == torch.bfloat16:
            x = Fn.to_bf16(x)
        whole = self._train_whole_model(x, data, dev)
        if whole is not None:
            return whole
        for i in range(self.num_layers):
            edge_index, e_id, size = data.batch_adjs[i]
            edge_index = edge_index.to(dev)
            conv, norm = self.convs[i][0], self.convs[i][1] if len(self.convs[i]) > 1 else None
            ea_all = data.all.edge_attr
            if isinstance(norm, BatchNorm) and Fn.sage_train_layer_supported(x, conv.lin_e, norm.module):
                # conv + norm + ReLU as one library call forward, one backward
                plan = plan_for(edge_index, x.size(0), size[1])
                in_place = plan.has_edge_rows and ea_all.is_cuda and ea_all.device == x.device and ea_all.dtype == torch.float32 \
                    and ea_all.dim() == 2 and ea_all.stride(1) == 1
                # a block whose plan carries edge_rows (the GPU block builder): edge_attr[e_id] (:215) is read in place
                ea = ea_all if in_place else _dev_f32(ea_all[e_id.to(ea_all.device)], dev)
                x = Fn.sage_train_layer(x, plan, ea, conv.lin_e, conv.lin_j, conv.lin_i, norm.module, scene_rows=in_place)
                continue
            ea = _dev_f32(ea_all[e_id.to(ea_all.device)], dev)
            x = conv((x, x[:size[1]]), ea, edge_index)
            x = self._norm_act(self.convs[i], x)
        if self.clf.model.decoder:
            x = self._decode(x)
        return x.float() if x.dtype == torch.bfloat16 else x

    def _train_whole_model(self, x, data, dev):
        """Training-mode forward through ONE library call (and one in the backward) when every layer qualifies for the composite
        entry points: fp32 rows, lin_e a Linear over <= 32 attributes, BatchNorm in training mode after every conv and inside the
        decoder.  Returns the logits, or None (the per-layer path then runs)."""
        built = self._train_spec(x, data, dev)
        if built is None:
            return None
        spec, tail = built
        h = Fn.static_train_model(x, spec)
        if tail is not None:
            h = Fn.linear2(h, tail.weight, bias=tail.bias, out_f32=True)
        return h

    def _train_spec(self, x, data, dev):
        """-> (per-layer spec for Fn.static_train_model / ops.static_train_fwd, the decoder's output Linear when it does NOT ride in the call | None),
        or None when a layer does not qualify (see _train_whole_model)"""
        from .. import ops
        if not (ops.TRAIN_COMPOSITE and ops.TRAIN_WHOLE_MODEL) or x.dtype != torch.float32 or self.num_layers + 2 > 8:
            return None
        # the module tree is read once per model (nn.Module.__getattr__ / Sequential.__getitem__ for every layer were a sixth of the step's host
        # time, round 6); `invalidate_caches()` after replacing a sub-module
        mods = self.__dict__.get("_train_mods")
        if mods is None:
            dec_ = tuple(self.decoder) if self.clf.model.decoder else ()
            mods = self.__dict__["_train_mods"] = (dec_, [(self.convs[i][0], self.convs[i][1] if len(self.convs[i]) > 1 else None) for i in range(self.num_layers)])
        dec, conv_mods = mods
        if len(dec) not in (0, 4) or (len(dec) == 4 and not isinstance(dec[1], BatchNorm)):
            return None
        ea_all = data.all.edge_attr
        spec, n_src = [], x.size(0)
        for i in range(self.num_layers):
            edge_index, e_id, size = data.batch_adjs[i]
            conv, norm = conv_mods[i]
            if not isinstance(norm, BatchNorm) or not Fn.sage_train_layer_supported(x, conv.lin_e, norm.module) or size[0] != n_src:
                return None
            plan = plan_for(edge_index.to(dev), size[0], size[1])
            in_place = plan.has_edge_rows and ea_all.is_cuda and ea_all.device == x.device and ea_all.dtype == torch.float32 \
                and ea_all.dim() == 2 and ea_all.stride(1) == 1
            ea = None
            if conv.lin_e is not None:
                ea = ea_all if in_place else _dev_f32(ea_all[e_id.to(ea_all.device)], dev)
            spec.append(dict(plan=plan, edge_attr=ea, scene_rows=in_place and conv.lin_e is not None, lin_e=conv.lin_e, lin_j=conv.lin_j, lin_i=conv.lin_i,
                             bn=norm.module))
            n_src = size[1]
        tail = None
        if len(dec) == 4:
            if not dec[1].module.training or dec[1].module.momentum is None:
                return None
            spec.append(dict(plan=None, n_rows=n_src, edge_attr=None, scene_rows=False, lin_e=None, lin_j=dec[0], lin_i=None, bn=dec[1].module))
            if ops.TRAIN_DECODER_OUTPUT_IN_CALL and isinstance(dec[3], torch.nn.Linear) and dec[3].bias is not None:
                # the decoder's output Linear rides in the same two library calls (a layer without BatchNorm / ReLU)
                spec.append(dict(plan=None, n_rows=n_src, edge_attr=None, scene_rows=False, lin_e=None, lin_j=dec[3], lin_i=None, bn=None))
            else:
                tail = dec[3]
        return spec, tail

    def train_step_direct(self, data, loss_fn):
        """Forward, loss and backward of ONE training step without the autograd engine (round 4; reference learning/runModel.py:266-279: forward,
        calcLossAndOA, loss.backward()): the whole-model library calls each way (dgnn_static_train_fwd / _bwd) are issued directly around
        `loss_fn(logits) -> (loss, dlogits)`, the parameters' .grad are SET to the fresh gradients (what zero_grad + backward leave).  Same kernels,
        same order as the autograd node (functional._StaticTrainModel): same numbers.  Returns the detached loss, or None when the model does not
        take the whole-model calls with logits out of the call (the caller then runs the autograd path)."""
        from .. import ops
        dev = self._device()
        x_all = data.all.x
        n_id = data.batch_n_id.to(x_all.device)
        x = _block_x(x_all, n_id, 1 if self.clf.regularization.cell_type else 0, dev)
        if self.storage_dtype != torch.float32:
            return None
        built = self._train_spec(x, data, dev)
        if built is None or built[1] is not None or len(built[0]) <= self.num_layers:
            return None
        spec = built[0]
        # what a layer's table holds of the MODEL (its parameters): read through the modules once, reused while the same modules come back
        static = self.__dict__.get("_direct_static")
        mods_now = tuple(id(sp[k]) for sp in spec for k in ("lin_e", "lin_j", "lin_i", "bn"))
        if static is None or static[0] != mods_now:
            tabs = []
            for sp in spec:
                le, lj, li, bn = sp["lin_e"], sp["lin_j"], sp["lin_i"], sp["bn"]
                tabs.append(dict(We=le.weight if le is not None else None, be=le.bias if le is not None else None, Wj=lj.weight, bj=lj.bias,
                                 Wi=li.weight if li is not None else None, gamma=bn.weight if bn is not None else None,
                                 beta=bn.bias if bn is not None else None, bn=bn))
            static = self.__dict__["_direct_static"] = (mods_now, tabs)
        layers = []
        for sp, st_ in zip(spec, static[1]):
            plan = sp["plan"]
            l = dict(st_)
            if plan is not None:
                l["plan_parts"], l["n_dst"], l["n_src"] = plan.part_ptrs(bool(sp["scene_rows"])), plan.n_dst, plan.n_src
            else:
                l["plan_parts"], l["n_dst"], l["n_src"] = None, sp["n_rows"], sp["n_rows"]
            l["edge_attr"] = sp["edge_attr"] if st_["We"] is not None else None
            layers.append(l)
        with torch.no_grad():
            logits, buf, meta = ops.static_train_fwd(x, layers)
            loss, dlogits = loss_fn(logits)
            for l, sp in zip(layers, spec):
                if sp["plan"] is not None:
                    l["t_parts"] = sp["plan"].transposed_ptrs(bool(sp["scene_rows"]))
            grads = ops.static_train_bwd(x, layers, buf, meta, dlogits, keep=self.__dict__.setdefault("_grad_keep", {}) if ops.TRAIN_KEEP_GRADS else None)
        written = set()
        for l, g in zip(layers, grads):
            for name, gr in zip(("We", "be", "Wj", "bj", "Wi", "gamma", "beta"), g):
                p_ = l[name]
                if p_ is not None and p_.requires_grad:
                    if p_.grad is not gr:       # (kept gradient tensors: assigned once, rewritten in place by every step)
                        p_.grad = gr
                    written.add(id(p_))
        plist = self.__dict__.get("_param_list")      # (nn.Module.parameters() walks the module tree: 50 us a step)
        if plist is None:
            plist = self.__dict__["_param_list"] = list(self.parameters())
        for p_ in plist:          # what zero_grad + backward leave: a trainable parameter this step did not reach has NO gradient
            if p_.requires_grad and p_.grad is not None and id(p_) not in written:       # (a stale one from an earlier autograd step would be applied by Adam)
                p_.grad = None
        return loss.detach()

    # ---- INFERENCE, whole graph (reference :323-355; the benchmarked path) ---------------------
    @torch.no_grad()
    def inference_layer(self, data_all, plan: GraphPlan = None):
        dev = self._device()
        x = _dev_f32(data_all.x, dev)
        x = x[:, 1:] if self.clf.regularization.cell_type else x
        x = self._storage_input(x)
        xe = _dev_f32(data_all.edge_attr, dev)
        xe = xe[:, 1:] if self.clf.regularization.edge_type else xe
        edge_index = data_all.edge_index.to(dev)
        one = self._infer_one_call(x, xe, edge_index, plan)
        if one is not None:
            return one
        if plan is None:
            # whole scene as processing/data.py delivers it: 4 adjacency rows per cell (verified on the device, any
            # other layout falls through to the generic builder)
            plan = plan_for(edge_index, x.size(0), x.size(0), hint=ops.PLAN_HINT_REFERENCE)
        if self.fuses_decoder(self.num_layers - 1) and self._fusable_rows(x, self.num_layers - 1):
            return self._eval_layers(x, x.size(0), xe, [plan] * self.num_layers, sorted_attr=True, decode=True)   # last launch writes the logits
        x = self._eval_layers(x, x.size(0), xe, [plan] * self.num_layers, sorted_attr=True)
        return self._eval_decoder(x)

    def _one_call_tables(self, x, xe):
        """What dgnn_static_infer_fwd / dgnn_static_infer_rings_fwd / dgnn_static_infer_partitioned_fwd take for this model: (layers, decoder, prepared,
        with_dec, cache) -- or None when this configuration runs layer by layer: other storage / widths / filters, a per-layer profiling hook, rows the
        fused kernels do not take.  `cache`: a dict the ops wrappers keep their argument tables (ctypes arrays of the tensors' addresses) in.
        Walking the module tree (nn.Module.__getattr__, Sequential.__getitem__: ~180 lookups) was 70 % of a 0.14 ms call on a reconbench-size scene,
        so the tables are kept until one of the tensors they were made from is written to or moved: the (address, version) pairs of the parameters
        and BatchNorm buffers -- read through the modules found on the first call -- are the key (load_state_dict, optimizer steps, .to(), train-mode
        running statistics all change it).  Replacing a sub-MODULE of a built model is not seen: call `invalidate_caches()` after such surgery."""
        if not (ops.INFER_ONE_CALL and ops.FUSED_ENABLED and ops.EDGE_GATHER_IN_KERNEL) or ops.LAYER_HOOK is not None:
            return None
        bf16 = self.storage_dtype == torch.bfloat16      # (the fully fused bf16-storage chain: dgnn_static_infer_rings_fwd_bf16)
        if (self.storage_dtype != torch.float32 and not bf16) or x.dtype != torch.float32 or xe.dtype != torch.float32 or xe.dim() != 2 or xe.size(1) != 20 \
                or xe.stride(0) != 20 or xe.data_ptr() % 16 or x.size(0) * max(x.stride(0), 128) >= ops.FUSED_MAX_ELEMS:
            return None
        if bf16 and (ops.BF16_MODE != ops.BF16_COMPENSATED or self.num_layers < 2):
            return None
        flags = (ops.GEMM_MODE, ops.PREPARED_PARAMS, ops.FUSE_DECODER, bf16, x.device)
        hit = self.__dict__.get("_oc_tables")
        if hit is not None:
            mods, key, tabs = hit
            if key == (self._oc_key(mods), flags):
                return tabs
        mods = []            # every module a tensor of the tables comes from
        dec = self.decoder if self.clf.model.decoder else ()
        if len(dec) not in (0, 4) or (len(dec) == 4 and not (isinstance(dec[0], nn.Linear) and isinstance(dec[3], nn.Linear)
                                                              and (dec[1] is None or isinstance(dec[1], BatchNorm)))):
            return None
        layers, prepared, last = [], [], self.num_layers - 1
        with_dec = len(dec) == 4 and self.fuses_decoder(last)
        for i, layer in enumerate(self.convs):
            conv = layer[0]
            le = conv.lin_e
            if not (isinstance(le, Linear) and le.in_features == 20 and le.bias is not None) or not isinstance(layer[2], nn.ReLU) \
                    or not (layer[1] is None or isinstance(layer[1], BatchNorm)):
                return None
            if bf16 and not ops.fused_layer_supported_bf16(conv.lin_j.in_features, conv.lin_j.out_features, 20, x if i == 0 else None):
                return None
            scale, shift = self._fold(layer[1], conv.lin_j.out_features, x.device)
            layers.append((le.weight, le.bias, conv.lin_j.weight, conv.lin_j.bias, conv.lin_i.weight, scale, shift))
            prepared.append(None if bf16 else self._prepared(i, with_dec and i == last))
            mods += [le, conv.lin_j, conv.lin_i] + ([layer[1].module] if layer[1] is not None else [])
        decoder = None
        if len(dec) == 4:
            s1, h1 = self._fold(dec[1], dec[0].out_features, x.device)
            decoder = (dec[0].weight, dec[0].bias, s1, h1, dec[3].weight, dec[3].bias)
            mods += [dec[0], dec[3]] + ([dec[1].module] if dec[1] is not None else [])
        if bf16 and not with_dec:          # (the bf16 chain has no decoder-apart form)
            return None
        tabs = (layers, decoder, None if bf16 else prepared, with_dec, {})
        self.__dict__["_oc_tables"] = (mods, (self._oc_key(mods), flags), tabs)
        return tabs

    @staticmethod
    def _oc_key(mods):
        key = []
        for m in mods:
            for t in m._parameters.values():
                if t is not None:
                    key.append(t.data_ptr())
                    key.append(t._version)
            for t in m._buffers.values():
                if t is not None:
                    key.append(t.data_ptr())
                    key.append(t._version)
        return key

    def invalidate_caches(self):
        """Forget everything derived from the parameters (folded BatchNorm, prepared layer blocks, the one-call tables): after replacing a sub-module."""
        for k in ("_oc_tables", "_fold_cache", "_prep_cache", "_train_mods", "_direct_static", "_param_list", "_grad_keep"):
            self.__dict__.pop(k, None)

    def _infer_one_call(self, x, xe, edge_index, plan):
        """The whole eval forward -- plan (unless the caller's or a cached one exists), every conv layer, the decoder -- as ONE library call
        (dgnn_static_infer_fwd issues the launches `_eval_layers` issues, in its order: bit-identical).  Returns the logits, or None when this
        configuration runs layer by layer (`_one_call_tables`)."""
        from ..graph import register_plan
        if edge_index.dtype != torch.int64:
            return None
        tabs = self._one_call_tables(x, xe)
        if tabs is None:
            return None
        layers, decoder, prepared, with_dec, cache = tabs
        n = x.size(0)
        if plan is None:
            held = getattr(edge_index, "_dgnn_plans", None)       # a resident scene: the plan of an earlier call (plan_for's cache)
            if held:
                plan = plan_for(edge_index, n, n, hint=ops.PLAN_HINT_REFERENCE)
        parts = None if plan is None else (plan.rowptr, plan.src, plan.eid)
        if self.storage_dtype == torch.bfloat16:
            out = ops.static_infer_rings_fwd_bf16(x, xe, edge_index, parts, [n] * self.num_layers, layers, decoder, hint=ops.PLAN_HINT_REFERENCE,
                                                  attr_in_plan_order=False, cache=cache)
        else:
            out = ops.static_infer_fwd(x, xe, edge_index, parts, layers, decoder, prepared, fuse_decoder=with_dec, cache=cache)
        if out is None:
            return None
        if plan is None:
            register_plan(edge_index, GraphPlan(edge_index, n, n, hint=ops.PLAN_HINT_REFERENCE, parts=out[1]))
        return out[0]

    def _fusable_rows(self, x, i):
        """the fused launches take packed 20-column fp32 edge rows and 16-byte aligned feature rows; other inputs run layer and decoder apart"""
        return (x.dtype == torch.float32 or self.storage_dtype == torch.bfloat16) and not self.clf.regularization.edge_type

    def _fold(self, norm, c, device):
        """BatchNorm(eval) as a per-channel (scale, shift) pair, cached until one of its tensors is written to
        (load_state_dict / optimizer steps / train-mode running statistics all bump the tensors' version counters)."""
        if norm is None:
            return None, None
        bn = norm.module
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t.data_ptr(), t._version) for t in ts) + (bn.eps,)
        cache = self.__dict__.setdefault("_fold_cache", {})
        hit = cache.get(id(bn))
        if hit is not None and hit[0] == key:
            return hit[1]
        out = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        cache[id(bn)] = (key, out)
        return out

    def _prepared(self, i, with_decoder):
        """Prepared parameters of conv layer i for the fused launches (ops.sage_layer_prepare), cached until one of the tensors they were made from is
        written to (same version-counter key as `_fold`); None when there is no prepared form (other arithmetic, other shapes, DGNN_PREPARED=0)."""
        if not ops.PREPARED_PARAMS or ops.GEMM_MODE != ops.GEMM_F16X2:
            return None
        conv = self.convs[i][0]
        le = conv.lin_e
        if not isinstance(le, Linear) or le.in_features != 20:
            return None
        ts = [le.weight, le.bias, conv.lin_j.weight, conv.lin_i.weight]
        dec_args = None
        if with_decoder:
            dec = self.decoder
            s1, h1 = self._fold(dec[1] if isinstance(dec[1], BatchNorm) else None, dec[0].out_features, le.weight.device)
            dec_args = (dec[0].weight, dec[0].bias, s1, h1, dec[3].weight, dec[3].bias)
            ts += [t for t in dec_args if t is not None]
        key = (bool(with_decoder),) + tuple((t.data_ptr(), t._version) for t in ts)
        cache = self.__dict__.setdefault("_prep_cache", {})
        hit = cache.get((i, bool(with_decoder)))
        if hit is not None and hit[0] == key:
            return hit[1]
        buf = ops.sage_layer_prepare(le.weight, le.bias, conv.lin_j.weight, conv.lin_i.weight, dec_args)
        cache[(i, bool(with_decoder))] = (key, buf)
        return buf

    def fuses_decoder(self, i):
        """True when layer i's launch also carries the decoder (the last conv layer of the shipped widths, fp32 storage or -- round 4 -- bf16 storage in
        the compensated arithmetic: the finished tile goes through Linear-BN-ReLU-Linear in the same kernel and only the logits are written, reference
        :180-187 applied at :350-351)"""
        if i != self.num_layers - 1 or not self.clf.model.decoder or len(self.decoder) != 4:
            return False
        conv, dec = self.convs[i][0], self.decoder
        le = conv.lin_e
        if not (isinstance(le, Linear) and isinstance(dec[0], nn.Linear) and isinstance(dec[3], nn.Linear) and isinstance(dec[2], nn.ReLU)):
            return False
        if not isinstance(self.convs[i][2], nn.ReLU) or (dec[1] is not None and not isinstance(dec[1], BatchNorm)):
            return False
        sup = ops.fused_layer_decoder_supported_bf16 if self.storage_dtype == torch.bfloat16 else ops.fused_layer_decoder_supported
        return sup(conv.lin_j.in_features, conv.lin_j.out_features, le.in_features, dec[0].out_features,
                   dec[3].out_features) and dec[0].in_features == conv.lin_j.out_features

    def _eval_layers(self, x, n_dst0, xe, plans, sorted_attr, only=None, out=None, rows=None, decode=False):
        """Eval-mode conv stack: per layer one fused launch when the widths allow it, else the
        aggregate + linear pair; BatchNorm(eval) and ReLU always ride in the GEMM epilogue.
        `sorted_attr`: True = xe rows follow the caller's edge_index order (gathered by eid in the kernel or staged once),
        False = xe rows are already in plan order.  `only=i` runs just layer i (the partitioned forward exchanges halos between layers), `out` then
        optionally names the [>= n_dst, C_out] buffer to write into and `rows=(b, e)` restricts the launch to
        the destinations [b, e) (written to out[b:e]; interior / boundary cells of a partition).  `decode`: when the last layer of the
        range `fuses_decoder`, its launch carries the decoder and LOGITS come back (`out`, if given, is then the [>= n_dst, 2] logits buffer)."""
        for i in (range(self.num_layers) if only is None else [only]):
            hook = ops.LAYER_HOOK
            conv_i = self.convs[i][0]
            n_rows = plans[i].n_dst if rows is None else rows[1] - rows[0]
            dec_i = bool(decode) and self.fuses_decoder(i)
            tok = hook(None, x.size(1), conv_i.lin_j.out_features, n_rows, not dec_i) if hook is not None else None
            x = self._eval_layer(i, x, xe, plans[i], sorted_attr, out, rows, dec_i)
            if hook is not None:
                hook(tok, conv_i.lin_i.in_features, conv_i.lin_j.out_features, n_rows, not dec_i)
        return x

    def _eval_layer(self, i, x, xe, plan, sorted_attr, out, rows, decode=False):
        """One eval-mode conv layer + BN + ReLU (see _eval_layers); `decode`: + the decoder, logits come back."""
        layer = self.convs[i]
        conv = layer[0]
        if decode:
            le_ = conv.lin_e
            if self.storage_dtype == torch.bfloat16:
                fusable = (isinstance(le_, Linear) and le_.in_features == 20
                           and ops.fused_layer_supported_bf16(conv.lin_j.in_features, conv.lin_j.out_features, 20, x)
                           and ops.fused_layer_decoder_supported_bf16(conv.lin_j.in_features, conv.lin_j.out_features, 20, self.decoder[0].out_features,
                                                                      self.decoder[3].out_features, x))
            else:
                fusable = (x.dtype == torch.float32 and isinstance(le_, Linear) and le_.in_features == 20
                           and ops.fused_layer_supported(x.size(1), conv.lin_j.out_features, 20, x)
                           and ops.fused_layer_decoder_supported(x.size(1), conv.lin_j.out_features, 20, self.decoder[0].out_features,
                                                                 self.decoder[3].out_features, x))
            if not fusable:     # this input cannot take the one-launch form (alignment, strides ...): layer and decoder apart, same interface
                lg = self._eval_decoder(self._eval_layer(i, x, xe, plan, sorted_attr, None, rows, False))
                if out is None:
                    return lg
                b_, e_ = (0, plan.n_dst) if rows is None else rows
                out[b_:e_] = lg
                return out
        norm = layer[1] if isinstance(layer[1], BatchNorm) else None
        scale, shift = self._fold(norm, conv.lin_j.out_features, x.device)
        le = conv.lin_e
        b, e = (0, plan.n_dst) if rows is None else rows
        n, rowptr = e - b, (plan.rowptr if rows is None else plan.rowptr[b:e + 1])
        x_dst = x[b:e]
        out_v = out if (out is None or rows is None) else out[b:e]
        simple = isinstance(le, Linear) and le.in_features in (2, 20)
        if x.dtype in (torch.bfloat16, ops.UROWS) or (self.storage_dtype == torch.bfloat16 and simple and le.in_features == 20
                                                      and ops.fused_layer_supported_bf16(conv.lin_j.in_features, conv.lin_j.out_features, 20, x)):
            c_in = conv.lin_j.in_features    # the logical width: bf16 rows may carry zero padding columns
            if not (simple and le.in_features == 20 and ops.fused_layer_supported_bf16(c_in, conv.lin_j.out_features, 20, x)):
                return self._eval_layer_bf16_unfused(conv, scale, shift, x, xe, plan, sorted_attr, out_v, rows)
            # row format of the output (ops.UROWS): a fused layer on fp32 feature rows starts the unsigned format, one on 16-bit rows keeps its input's;
            # a caller-supplied buffer names the format by its dtype
            # (the decoder-carrying launch writes fp32 logits: no row format to agree on)
            uns = (ops.BF16_UNSIGNED_ROWS and ops.BF16_MODE == ops.BF16_COMPENSATED and x.dtype != torch.bfloat16) if (out_v is None or decode) \
                else out_v.dtype == ops.UROWS
            if not decode and x.dtype != torch.float32 and uns != (x.dtype == ops.UROWS):
                raise ops.DgnnError("bf16 storage: layer %d reads %s rows but is asked to write %s rows" % (i, x.dtype, out_v.dtype))
            if sorted_attr and ops.EDGE_GATHER_IN_KERNEL and xe.stride(0) == 20 and xe.data_ptr() % 16 == 0:
                ea, eid = xe, plan.eid
            else:
                ea, eid = (plan.sorted_edge_attr(xe) if sorted_attr else xe), None
                if ea.stride(0) != 20 or ea.data_ptr() % 16:
                    ea = ea.contiguous()
            if decode:      # (`fusable` above: the launch carries the decoder and writes fp32 logits)
                dec = self.decoder
                s1, h1 = self._fold(dec[1] if isinstance(dec[1], BatchNorm) else None, dec[0].out_features, x.device)
                return ops.sage_layer_fused_decoder_fwd_bf16(rowptr, plan.src, n, x, c_in, ea, le.weight, le.bias, conv.lin_j.weight, conv.lin_j.bias,
                                                             conv.lin_i.weight, scale, shift, True, dec[0].weight, dec[0].bias, s1, h1, dec[3].weight,
                                                             dec[3].bias, out=out_v, eid=eid, x_dst=x_dst if b else None)
            return ops.sage_layer_fused_fwd_bf16(rowptr, plan.src, n, x, c_in, ea, le.weight, le.bias, conv.lin_j.weight, conv.lin_j.bias,
                                                 conv.lin_i.weight, scale, shift, True, out=out_v, eid=eid, x_dst=x_dst if b else None,
                                                 rows_out_unsigned=uns)
        if simple and le.in_features == 20 and ops.fused_layer_supported(x.size(1), conv.lin_j.out_features, 20, x):
            # sorted_attr: xe is in the caller's edge order.  Either the kernel gathers each row by eid (no staging
            # copy of the edge features), or the rows are staged into plan order once and reused by all layers.
            if sorted_attr and ops.EDGE_GATHER_IN_KERNEL and xe.stride(0) == 20 and xe.data_ptr() % 16 == 0:
                ea, eid = xe, plan.eid
            else:
                ea, eid = (plan.sorted_edge_attr(xe) if sorted_attr else xe), None
                if ea.stride(0) != 20 or ea.data_ptr() % 16:
                    ea = ea.contiguous()  # e.g. a [:, 1:] view of 21-column rows (regularization.edge_type)
            # the kernels address rows with 32-bit element offsets relative to x_dst: beyond 2^31 elements per launch
            # (16.7M cells at 128 channels) the destinations are processed as consecutive sub-ranges
            if x.size(0) * x.stride(0) >= (1 << 32):
                raise ops.DgnnError("fused layer: source rows beyond 2^32 elements (%d x %d); partition the scene "
                                    "(dgnn_amd.partition)" % (x.size(0), x.stride(0)))
            chunk = max(1, (ops.FUSED_MAX_ELEMS - 1) // max(x.stride(0), 1))
            if decode and ops.fused_layer_decoder_supported(x.size(1), conv.lin_j.out_features, 20, self.decoder[0].out_features,
                                                            self.decoder[3].out_features, x):
                dec = self.decoder
                s1, h1 = self._fold(dec[1] if isinstance(dec[1], BatchNorm) else None, dec[0].out_features, x.device)
                if out_v is None:
                    out_v = torch.empty((n, 2), dtype=torch.float32, device=x.device)
                for s0 in range(0, n, chunk):
                    s1_ = min(n, s0 + chunk)
                    ops.sage_layer_fused_decoder_fwd(rowptr[s0:s1_ + 1] if (s0 or s1_ < n) else rowptr, plan.src, s1_ - s0, x, ea, le.weight, le.bias,
                                                     conv.lin_j.weight, conv.lin_j.bias, conv.lin_i.weight, scale, shift, True, dec[0].weight, dec[0].bias,
                                                     s1, h1, dec[3].weight, dec[3].bias, out=out_v[s0:s1_], eid=eid,
                                                     x_dst=x_dst[s0:s1_] if (b or s0) else None, prepared=self._prepared(i, True))
                return out_v
            if decode:
                raise ops.DgnnError("decode=True on a layer whose launch cannot carry the decoder (check fuses_decoder first)")
            if n <= chunk:
                x = ops.sage_layer_fused_fwd(rowptr, plan.src, n, x, ea, le.weight, le.bias, conv.lin_j.weight,
                                             conv.lin_j.bias, conv.lin_i.weight, scale, shift, True, out=out_v, eid=eid,
                                             x_dst=x_dst if b else None, prepared=self._prepared(i, False))
            else:
                if out_v is None:
                    out_v = torch.empty((n, conv.lin_j.out_features), dtype=torch.float32, device=x.device)
                for s0 in range(0, n, chunk):
                    s1 = min(n, s0 + chunk)
                    ops.sage_layer_fused_fwd(rowptr[s0:s1 + 1], plan.src, s1 - s0, x, ea, le.weight, le.bias, conv.lin_j.weight,
                                             conv.lin_j.bias, conv.lin_i.weight, scale, shift, True, out=out_v[s0:s1], eid=eid,
                                             x_dst=x_dst[s0:s1], prepared=self._prepared(i, False))
                x = out_v
            return x
        if simple and le.in_features == 20 and out is None and rows is None and self.wide_layer_split_rows(x.size(1), conv.lin_j.out_features):
            y = self._eval_layer_wide(i, conv, scale, shift, x, xe, plan, sorted_attr)
            if y is not None:
                return y
        if isinstance(x, ops.SplitRows):       # a layer the wide kernels do not take behind one they did: back to fp32 rows
            x = x.float()
            x_dst = x[b:e]
        if simple:
            ea = plan.sorted_edge_attr(xe) if sorted_attr else xe
            a = ops.aggregate_fwd(rowptr, plan.src, None, n, x, ea, le.weight, le.bias)
        else:
            if rows is not None:
                raise NotImplementedError("destination sub-ranges need a Linear edge filter")
            a = Fn.aggregate(x, plan, **conv._filter_args(xe))
        return ops.linear_fwd(a, conv.lin_j.weight, x_dst, conv.lin_i.weight, conv.lin_j.bias, scale, shift, True, out=out_v)

    # ---- wide conv layers on split rows (round 5; csrc/wide.hip) -------------------------------------------------------------------------------
    def wide_layer_split_rows(self, c_in, c_out):
        """True when the eval-mode conv layer c_in -> c_out runs on dgnn_sage_aggregate_sr + dgnn_linear_sr (fp32 storage, default arithmetic, the widths
        of the reference's real configs: configs/eth.yaml:56, aerial.yaml:57, modelnet.yaml:56)"""
        return self.storage_dtype == torch.float32 and ops.wide_layer_supported(int(c_in), int(c_out), 20)

    def _wide_prepared(self, key, tensors, make):
        """what the wide kernels derive from the weights (split weight rows, the filter operand), cached until one of `tensors` is written to"""
        k = tuple((t.data_ptr(), t._version) for t in tensors)
        cache = self.__dict__.setdefault("_prep_cache", {})
        hit = cache.get(("wide",) + tuple(key))
        if hit is not None and hit[0] == k:
            return hit[1]
        val = make()
        cache[("wide",) + tuple(key)] = (k, val)
        return val

    def _eval_layer_wide(self, i, conv, scale, shift, x, xe, plan, sorted_attr):
        """conv + BatchNorm(eval) + ReLU of a wide layer: the mean aggregate and the own rows as SPLIT ROWS (ops.SplitRows), the dense product straight on
        them, the output again split rows -- what the next wide layer and the decoder read.  None: the library declined the layout."""
        le = conv.lin_e
        if sorted_attr and ops.EDGE_GATHER_IN_KERNEL and xe.stride(0) == 20 and xe.data_ptr() % 16 == 0:
            ea, eid = xe, plan.eid
        else:
            ea, eid = (plan.sorted_edge_attr(xe) if sorted_attr else xe), None
            if ea.stride(0) != 20 or ea.data_ptr() % 16:
                ea = ea.contiguous()
        prep = self._wide_prepared((i, "filter"), (le.weight, le.bias), lambda: ops.sr_prepare_filter(le.weight, le.bias))
        Wp = self._wide_prepared((i, "dense"), (conv.lin_j.weight, conv.lin_i.weight), lambda: ops.pack_rows(conv.lin_j.weight, conv.lin_i.weight, per_row=True))
        if prep is None:
            return None
        n = plan.n_dst
        if isinstance(x, ops.SplitRows):
            a = ops.aggregate_sr(plan.rowptr, plan.src, eid, n, x, ea, le.weight, le.bias, prep)
            xi = x[:n]
        else:
            if x.stride(0) % 4 or x.data_ptr() % 16:
                x = x.contiguous()
            got = ops.aggregate_sr(plan.rowptr, plan.src, eid, n, x, ea, le.weight, le.bias, prep, own_rows=True)
            a, xi = got if got is not None else (None, None)
        if a is None:
            return None
        return ops.linear_sr(a, Wp, xi, conv.lin_j.bias, scale, shift, relu=True)

    def _eval_layer_bf16_unfused(self, conv, scale, shift, x, xe, plan, sorted_attr, out_v, rows):
        """bf16 storage, widths the fused bf16 kernel does not cover: the generic bf16 aggregate + bf16-MFMA GEMM pair."""
        if rows is not None:
            raise NotImplementedError("destination sub-ranges in bf16 storage need a fused-kernel width")
        le = conv.lin_e
        c_in = conv.lin_j.in_features
        if x.dtype == ops.UROWS:      # unsigned rows of a fused layer: the generic kernels take plain bf16
            x = ops.rows_unsigned_to_bf16(x)
        xs = x[:, :c_in] if x.size(1) != c_in else x
        if isinstance(le, Linear) and le.in_features in (2, 20):
            ea = plan.sorted_edge_attr(xe) if sorted_attr else xe
            a = ops.aggregate_fwd(plan.rowptr, plan.src, None, plan.n_dst, xs, ea, le.weight, le.bias)
        else:
            a = Fn.aggregate(xs, plan, **conv._filter_args(xe, True))
        return ops.linear_fwd(a, conv.lin_j.weight, xs[:plan.n_dst], conv.lin_i.weight, conv.lin_j.bias, scale, shift, True, out=out_v,
                              out_dtype=torch.bfloat16)

    def _eval_decoder(self, x):
        dec = self.decoder
        if isinstance(x, ops.SplitRows):
            # behind a wide layer: Linear(h3 -> h3/2) + BN + ReLU straight on the split rows (fp32 rows out), then the small output Linear
            if self.clf.model.decoder and len(dec) == 4 and isinstance(dec[0], nn.Linear) and dec[0].out_features % 256 == 0 and isinstance(dec[2], nn.ReLU) \
                    and (dec[1] is None or isinstance(dec[1], BatchNorm)):
                scale, shift = self._fold(dec[1] if isinstance(dec[1], BatchNorm) else None, dec[0].out_features, x.device)
                Wp = self._wide_prepared(("dec0",), (dec[0].weight,), lambda: ops.pack_rows(dec[0].weight, per_row=True))
                if isinstance(dec[3], nn.Linear) and dec[3].out_features in (1, 2):      # the output Linear rides in the same launch: only logits leave it
                    lg = ops.linear_sr(x, Wp, None, dec[0].bias, scale, shift, relu=True, proj=(dec[3].weight, dec[3].bias))
                    if lg is not None:
                        return lg
                h = ops.linear_sr(x, Wp, None, dec[0].bias, scale, shift, relu=True, out_f32=True)
                if h is not None:
                    return ops.linear_fwd(h, dec[3].weight, bias=dec[3].bias)
            x = x.float()
        if x.dtype == ops.UROWS:          # (a last layer that did not carry the decoder left unsigned rows)
            x = ops.rows_unsigned_to_bf16(x)
        if not self.clf.model.decoder or len(dec) == 0:
            return x.float() if x.dtype == torch.bfloat16 else x
        if len(dec) == 1:
            return ops.linear_fwd(x, dec[0].weight, bias=dec[0].bias, out_dtype=torch.float32)
        scale, shift = self._fold(dec[1] if isinstance(dec[1], BatchNorm) else None, dec[0].out_features, x.device)
        if x.dtype == torch.bfloat16:
            if ops.decoder_fused_supported(dec[0].in_features, dec[0].out_features, dec[3].out_features) and x.stride(0) % 8 == 0:
                return ops.decoder_fused_fwd_bf16(x, dec[0].weight, dec[0].bias, scale, shift, dec[3].weight, dec[3].bias)
            h = ops.linear_fwd(x, dec[0].weight, bias=dec[0].bias, scale=scale, shift=shift, relu=True, out_dtype=torch.bfloat16)
            return ops.linear_fwd(h, dec[3].weight, bias=dec[3].bias, out_dtype=torch.float32)
        if ops.decoder_fused_supported(dec[0].in_features, dec[0].out_features, dec[3].out_features):
            return ops.decoder_fused_fwd(x, dec[0].weight, dec[0].bias, scale, shift, dec[3].weight, dec[3].bias)
        h = ops.linear_fwd(x, dec[0].weight, bias=dec[0].bias, scale=scale, shift=shift, relu=True)
        return ops.linear_fwd(h, dec[3].weight, bias=dec[3].bias)

    # ---- INFERENCE, batch-major k-hop blocks (reference :232-275) ------------------------------
    @torch.no_grad()
    def inference_batch_layer(self, data_all, batch_loader):
        dev = self._device()
        x_out = torch.zeros([data_all.x.size(0), 2 if self.clf.training.loss == "kl" else 1], dtype=torch.float32, device=dev)
        x_all = _dev_f32(data_all.x, dev)
        x_all = x_all[:, 1:] if self.clf.regularization.cell_type else x_all
        xe_all = _dev_f32(data_all.edge_attr, dev)
        xe_all = xe_all[:, 1:] if self.clf.regularization.edge_type else xe_all
        for batch_size, n_id, adjs in batch_loader:
            n_id = n_id.to(dev)
            x = ops.gather_rows(x_all, n_id.to(torch.int32))
            for i in range(self.num_layers):
                edge_index, e_id, size = adjs[i]
                # blocks from dgnn_amd.sampler come with their plan registered (a cache hit); any other producer's
                # block is sorted here.  The fused layer then reads the block's edge rows through the plan's eid.
                plan = plan_for(edge_index.to(dev), size[0], size[1], hint=ops.PLAN_HINT_GROUPED)
                ea = ops.gather_rows(xe_all, e_id.to(dev).to(torch.int32))
                x = self._eval_layers(x, plan.n_dst, ea, [plan] * self.num_layers, True, only=i)
            x = self._eval_decoder(x)
            ops.scatter_rows_(x_out, n_id[:batch_size], x)
        return x_out

    def _eval_layers_one(self, i, x, ea, plan):
        """Layer i as the unfused aggregate + GEMM pair (any width; the row-level tests use it as the reference path)."""
        layer = self.convs[i]
        conv = layer[0]
        norm = layer[1] if isinstance(layer[1], BatchNorm) else None
        scale, shift = self._fold(norm, conv.lin_j.out_features, x.device)
        le = conv.lin_e
        if isinstance(le, Linear) and le.in_features in (2, 20):
            a = ops.aggregate_fwd(plan.rowptr, plan.src, plan.eid, plan.n_dst, x, ea, le.weight, le.bias)
        else:
            a = Fn.aggregate(x, plan, **conv._filter_args(ea))
        return ops.linear_fwd(a, conv.lin_j.weight, x[:plan.n_dst], conv.lin_i.weight, conv.lin_j.bias, scale, shift, True)

    # ---- INFERENCE, layer-major 1-hop blocks (reference :279-320) -------------------------------
    @torch.no_grad()
    def inference_layer_batch(self, data_all, batch_loader):
        dev = self._device()
        x_all = _dev_f32(data_all.x, dev)
        x_all = x_all[:, 1:] if self.clf.regularization.cell_type else x_all
        xe_all = _dev_f32(data_all.edge_attr, dev)
        xe_all = xe_all[:, 1:] if self.clf.regularization.edge_type else xe_all
        for i in range(self.num_layers):
            xs = []
            for batch_size, n_id, adj in batch_loader:
                edge_index, e_id, size = adj
                x = ops.gather_rows(x_all, n_id.to(dev).to(torch.int32))
                plan = plan_for(edge_index.to(dev), size[0], size[1], hint=ops.PLAN_HINT_GROUPED)
                ea = ops.gather_rows(xe_all, e_id.to(dev).to(torch.int32))
                y = self._eval_layers(x, plan.n_dst, ea, [plan] * self.num_layers, True, only=i)   # stays in HBM
                # a >= 256-wide layer hands back split rows (ops.SplitRows); this schedule concatenates the batches and gathers the next
                # layer's inputs by row, so they go back to fp32 rows here (exact: hi + lo is the stored value)
                xs.append(y.float() if isinstance(y, ops.SplitRows) else y)
            x_all = torch.cat(xs, dim=0)
        return self._eval_decoder(x_all)
