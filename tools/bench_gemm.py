"""GEMM micro-benchmark: dgnn_linear_fwd (bit-faithful fp32 MFMA) vs dgnn_linear_fwd_x3 (exact 3-way bf16 split, 6 products) vs
dgnn_linear_fwd_bf16, on the shapes of the wide conv layers and of the training step.  Prints fp32-equivalent TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgnn_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [(1010078, 128, 128, 256), (1010078, 256, 256, 512), (1010078, 512, 512, 1024), (1010078, 512, 0, 256), (138000, 128, 128, 128),
          (138000, 64, 64, 128), (552000, 128, 0, 128)]
for M, k1, k2, n in shapes:
    A1, W1 = torch.randn(M, k1, device=dev, generator=g), torch.randn(n, k1, device=dev, generator=g) * 0.1
    A2, W2 = (torch.randn(M, k2, device=dev, generator=g), torch.randn(n, k2, device=dev, generator=g) * 0.1) if k2 else (None, None)
    b = torch.randn(n, device=dev, generator=g)
    flop = 2.0 * M * (k1 + k2) * n
    ref = None
    line = "M=%8d K=%3d+%3d N=%4d:" % (M, k1, k2, n)
    for name, mode in (("f32", ops.GEMM_F32), ("x3", ops.GEMM_BF16X3_FILTER), ("x2h", ops.GEMM_F16X2)):
        ops.GEMM_MODE = mode
        f = lambda: ops.linear_fwd(A1, W1, A2, W2, b, relu=True)
        out = f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        if ref is None:
            ref = out
            err = 0.0
        else:
            err = ((out - ref).abs().max() / ref.abs().max()).item()
        line += "  %s %.3f ms %.0f TF (rel diff %.1e)" % (name, ms, flop / ms / 1e9, err)
    A1b, A2b = ops.cast_to_bf16(A1), (ops.cast_to_bf16(A2) if k2 else None)
    f = lambda: ops.linear_fwd(A1b, W1, A2b, W2, b, relu=True)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    line += "  bf16 %.3f ms %.0f TF" % (ms, flop / ms / 1e9)
    print(line, flush=True)
    del A1, A2, A1b, A2b, out, ref
