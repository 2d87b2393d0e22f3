"""Ceiling reference for the encoder GEMM shapes: the rate the vendor library (torch.matmul -> hipBLASLt /
rocBLAS, assembly kernels) reaches on the same bf16 shapes, on random operands.  Not part of the product:
the engine calls no library GEMM.  Prints one line per shape."""
import sys
import torch

windows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = windows * 1500
dev = torch.device("cuda:0")
shapes = [("qkv", M, 3840, 1280), ("out", M, 1280, 1280), ("fc1", M, 5120, 1280), ("fc2", M, 1280, 5120)]
for name, m, n, k in shapes:
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    reps = 10
    for _ in range(reps):
        c = a @ w.t()
    end.record()
    torch.cuda.synchronize()
    ms = beg.elapsed_time(end) / reps
    print("%s  M=%d N=%d K=%d  %.3f ms  %.0f TFLOP/s" % (name, m, n, k, ms, 2.0 * m * n * k / ms / 1e9), flush=True)
    del a, w, c
