import os, sys
sys.path.insert(0, os.getcwd())
import dgnn_amd._lib as L
if len(sys.argv) > 1: L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from dgnn_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for M, k1, k2, n in [(1010078, 256, 256, 512), (1010078, 512, 512, 1024), (1010078, 512, 0, 512)]:
    A1, W1 = torch.randn(M, k1, device=dev, generator=g), torch.randn(n, k1, device=dev, generator=g) * 0.1
    A2, W2 = (torch.randn(M, k2, device=dev, generator=g), torch.randn(n, k2, device=dev, generator=g) * 0.1) if k2 else (None, None)
    b = torch.randn(n, device=dev, generator=g)
    flop = 2.0 * M * (k1 + k2) * n
    line = "M=%d K=%d+%d N=%d:" % (M, k1, k2, n)
    for rep in range(2):
      for name, mode in (("x3", ops.GEMM_BF16X3_FILTER), ("x2h", ops.GEMM_F16X2)):
        ops.GEMM_MODE = mode
        f = lambda: ops.linear_fwd(A1, W1, A2, W2, b, relu=True)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        line += "  %s %.3f ms %.0f TF" % (name, ms, flop / ms / 1e9)
    print(line, flush=True)
