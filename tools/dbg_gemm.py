import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgnn_amd import ops
dev = "cuda:0"
ops.GEMM_MODE = ops.GEMM_BF16X3
M, n = 1010078, 256
def t(f, it=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
A = torch.randn(M, 256, device=dev); A1 = torch.randn(M, 128, device=dev); A2 = torch.randn(M, 128, device=dev)
W = torch.randn(n, 256, device=dev) * 0.1; W1 = W[:, :128].contiguous(); W2 = W[:, 128:].contiguous()
out = torch.empty(M, n, device=dev)
print("K=256 single operand          %.3f ms" % t(lambda: ops.linear_fwd(A, W, out=out)))
print("K=128+128 two tensors         %.3f ms" % t(lambda: ops.linear_fwd(A1, W1, A2, W2, out=out)))
print("K=128+128 same tensor twice   %.3f ms" % t(lambda: ops.linear_fwd(A1, W1, A1, W2, out=out)))
print("K=128+128 views of one [M,256] %.3f ms" % t(lambda: ops.linear_fwd(A[:, :128], W1, A[:, 128:], W2, out=out)))
print("K=128 only                    %.3f ms" % t(lambda: ops.linear_fwd(A1, W1, out=out)))
print("K=128+128 two tensors, no out= %.3f ms" % t(lambda: ops.linear_fwd(A1, W1, A2, W2)))
