import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgnn_amd._lib as L
if os.environ.get("DGNN_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["DGNN_LIB"])
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
for k, n in ((256, 512),):
    A1 = torch.randn(M, k, device=dev); A2 = torch.randn(M, k, device=dev)
    W1 = torch.randn(n, k, device=dev) * 0.1; W2 = torch.randn(n, k, device=dev) * 0.1
    out = torch.empty(M, n, device=dev)
    ms = t(lambda: ops.linear_fwd(A1, W1, A2, W2, out=out))
    print("%s M=%d K=%d+%d N=%d: %.3f ms" % (os.environ.get("DGNN_LIB", "default"), M, k, k, n, ms))
