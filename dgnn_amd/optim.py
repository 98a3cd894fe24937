"""Adam for the training step of the drop-in models (reference learning/runModel.py:290 `torch.optim.Adam(self.model.parameters(), lr=...)`,
stepped at :282): same update rule and state layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), ONE library launch per step
(csrc/adam.hip) and a host side that reads 34 gradient addresses instead of walking torch's grouping / dispatch machinery (~0.1 ms of a 0.9 ms step).
fp32 CUDA parameters only; anything else raises at construction and the caller keeps torch.optim.Adam."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, lib, stream_ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        for group in self.param_groups:
            for p in group["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise TypeError("dgnn_amd.optim.Adam steps contiguous fp32 CUDA parameters (got %s on %s)" % (p.dtype, p.device))
            self._check_one_device(group["params"])
        self._tables = {}       # per group: (parameters with a gradient last time, pointer tables)

    @staticmethod
    def _check_one_device(ps):
        """one launch steps a whole group: its tensors must live on ONE GPU (torch.optim.Adam groups by device internally; the reference has one)"""
        devs = {p.device for p in ps}
        if len(devs) > 1:
            raise ValueError("dgnn_amd.optim.Adam: a parameter group spans devices %s; give every device its own group" % sorted(map(str, devs)))

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        g = self.param_groups[-1]
        for p in g["params"]:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise TypeError("dgnn_amd.optim.Adam steps contiguous fp32 CUDA parameters (got %s on %s)" % (p.dtype, p.device))
        self._check_one_device(g["params"])
        if hasattr(self, "_tables"):
            self._tables.clear()

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict replaces `self.state`: the cached pointer tables would keep stepping the OLD moment buffers.  A
        dict saved by torch.optim.Adam carries `step` as a tensor and moments of any layout: coerced to what the launch takes."""
        super().load_state_dict(state_dict)
        self._tables.clear()
        for p, st in self.state.items():
            if "step" in st:
                st["step"] = int(st["step"].item()) if isinstance(st["step"], torch.Tensor) else int(st["step"])
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()

    @staticmethod
    def supports(params) -> bool:
        return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params)

    def _init_state(self, ps):
        """moments of the parameters seen for the first time: views of one zero-filled buffer"""
        new = [p for p in ps if len(self.state[p]) == 0]
        if not new:
            return
        flat = torch.zeros(2 * sum(p.numel() for p in new), dtype=torch.float32, device=new[0].device)
        off = 0
        for p in new:
            n = p.numel()
            st = self.state[p]
            st["step"] = 0
            st["exp_avg"] = flat[off:off + n].view_as(p)
            st["exp_avg_sq"] = flat[off + n:off + 2 * n].view_as(p)
            off += 2 * n

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = lib()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            # kernels, the stream handle and the moment buffers follow the PARAMETERS' device, not the thread's current one (the reference addresses
            # its GPU as "cuda:<n>" without set_device, run.py:98,127; torch.optim.Adam, which this replaces, handles that case)
            dev = ps[0].device
            if dev.index != torch._C._cuda_getDevice():
                with torch.cuda.device(dev):
                    self._step_group(L, gi, group, ps)
            else:
                self._step_group(L, gi, group, ps)
        return loss

    def _step_group(self, L, gi, group, ps):
        tab = self._tables.get(gi)
        if tab is None or len(tab[0]) != len(ps) or any(a is not b for a, b in zip(tab[0], ps)) or any(a != b.data_ptr() for a, b in zip(tab[5], ps)):
            self._init_state(ps)
            n = len(ps)
            tab = (ps, (C.c_void_p * n)(*[p.data_ptr() for p in ps]), (C.c_void_p * n)(*[self.state[p]["exp_avg"].data_ptr() for p in ps]),
                   (C.c_void_p * n)(*[self.state[p]["exp_avg_sq"].data_ptr() for p in ps]), (C.c_int64 * n)(*[p.numel() for p in ps]),
                   [p.data_ptr() for p in ps], [self.state[p] for p in ps])
            self._tables[gi] = tab
        _, p_arr, m_arr, v_arr, n_arr, _, states = tab
        grads = []
        for p in ps:
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
                g = p.grad = g.to_dense().float().contiguous() if g.is_sparse else g.float().contiguous()
            grads.append(g.data_ptr())
        g_arr = (C.c_void_p * len(ps))(*grads)
        steps = {st["step"] for st in states}
        beta1, beta2 = group["betas"]
        if len(steps) == 1:
            t = states[0]["step"] + 1
            check(L.dgnn_adam_step(len(ps), p_arr, g_arr, m_arr, v_arr, n_arr, float(group["lr"]), beta1, beta2, group["eps"], t, stream_ptr()), "dgnn_adam_step")
            for st in states:
                st["step"] = t
            torch.autograd.graph.increment_version(ps)     # the launch wrote the parameters behind torch's back: caches keyed on _version must see it
        else:   # parameters that skipped steps (no gradient then) carry their own step count: one launch per count
            for t0 in sorted(steps):
                idx = [i for i, st in enumerate(states) if st["step"] == t0]
                sub = lambda arr, ty: (ty * len(idx))(*[arr[i] for i in idx])
                check(L.dgnn_adam_step(len(idx), sub(p_arr, C.c_void_p), sub(g_arr, C.c_void_p), sub(m_arr, C.c_void_p), sub(v_arr, C.c_void_p),
                                       sub(n_arr, C.c_int64), float(group["lr"]), beta1, beta2, group["eps"], t0 + 1, stream_ptr()), "dgnn_adam_step")
                for i in idx:
                    states[i]["step"] = t0 + 1
            torch.autograd.graph.increment_version(ps)
