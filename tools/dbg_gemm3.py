import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgnn_amd import ops
dev = "cuda:0"
ops.GEMM_MODE = ops.GEMM_BF16X3
def t(f, it=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
M = 1010078
g = torch.Generator(device=dev).manual_seed(0)
for k, k2, n in ((128, 128, 256), (256, 256, 512), (512, 512, 1024), (512, 0, 256), (70, 33, 300)):
    A1 = torch.randn(M, k, device=dev, generator=g); A2 = torch.randn(M, k2, device=dev, generator=g) if k2 else None
    W1 = torch.randn(n, k, device=dev, generator=g) * 0.1; W2 = torch.randn(n, k2, device=dev, generator=g) * 0.1 if k2 else None
    b = torch.randn(n, device=dev, generator=g)
    out = torch.empty(M, n, device=dev)
    ms = t(lambda: ops.linear_fwd(A1, W1, A2, W2, b, relu=True, out=out))
    chk = float(out.double().sum().item()), float(out[::997].abs().double().sum().item())
    print("M=%d K=%d+%d N=%d: %.3f ms  %.0f TFLOP/s  checksum %.6f %.6f" % (M, k, k2, n, ms, 2.0 * M * (k + k2) * n / ms / 1e9, chk[0], chk[1]))
    del A1, A2, out
