"""The training step's GEMM shapes at the reference's widths (configs/modelnet.yaml:44,56: [128,256,512,1024], batch 1024; shipped [64,128,128,128],
batch 2048): time of ops.linear_fwd per shape, x3 (fp32 storage) and bf16 storage.  Run once per kernel selection:
    DGNN_X3_SMALL=0 DGNN_BF16_SMALL=0 python tools/bench_gemm_train_shapes.py      (tiled kernels everywhere)
    DGNN_SMALL_SPLITK=0 python tools/bench_gemm_train_shapes.py                     (small kernels without split-K)
    python tools/bench_gemm_train_shapes.py                                          (the library's own choice)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgnn_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
ops.GEMM_MODE = ops.GEMM_BF16X3
shapes = [(50000, 128, 128, 256), (20000, 128, 128, 256), (20000, 256, 0, 256), (6000, 256, 256, 512), (6000, 512, 0, 512), (1024, 512, 512, 1024),
          (1024, 1024, 0, 1024), (1024, 1024, 0, 512), (1024, 512, 0, 1024), (2048, 128, 0, 64), (2048, 128, 128, 128), (15000, 128, 128, 128),
          (15000, 128, 0, 256), (3000, 512, 512, 1024), (12000, 512, 512, 1024)]
tag = "X3_SMALL=%s BF16_SMALL=%s SPLITK=%s MID=%s" % (os.environ.get("DGNN_X3_SMALL", "1"), os.environ.get("DGNN_BF16_SMALL", "1"), os.environ.get("DGNN_SMALL_SPLITK", "1"),
                                                 os.environ.get("DGNN_GEMM_MID", "1"))
print(tag)
for M, k1, k2, n in shapes:
    A1, W1 = torch.randn(M, k1, device=dev, generator=g), torch.randn(n, k1, device=dev, generator=g) * 0.1
    A2, W2 = (torch.randn(M, k2, device=dev, generator=g), torch.randn(n, k2, device=dev, generator=g) * 0.1) if k2 else (None, None)
    b = torch.randn(n, device=dev, generator=g)
    res = []
    for name in ("x3", "bf16"):
        if name == "bf16":
            a1, a2 = ops.cast_to_bf16(A1), (ops.cast_to_bf16(A2) if k2 else None)
        else:
            a1, a2 = A1, A2
        f = lambda: ops.linear_fwd(a1, W1, a2, W2, b, relu=True)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append("%s %6.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3))
    print("M=%6d K=%4d+%4d N=%4d:  %s" % (M, k1, k2, n, "   ".join(res)), flush=True)
