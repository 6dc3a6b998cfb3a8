"""Token-resident linear kernels: time against the number of output features at a fixed token count (separates the
per-tile cost from launch + prologue + epilogue), hipGraph replay of 10 back-to-back launches.

    python benchmarks/token_linear_sweep.py [tokens]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as F

T = int(sys.argv[1]) if len(sys.argv) > 1 else 22726


def graph_time(fn, reps=10):
    with torch.no_grad():
        for _ in range(3):
            fn()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


torch.manual_seed(0)
x = torch.randn(2, T // 2, 256, device="cuda").to(torch.bfloat16)
pos = torch.randn(2, T // 2, 256, device="cuda").to(torch.bfloat16)
norm = torch.nn.LayerNorm(256).cuda().to(torch.bfloat16)
for N in (32, 128, 256, 384, 768, 1536):
    lin = torch.nn.Linear(256, N).cuda().to(torch.bfloat16)
    a = graph_time(lambda: F.token_linear(x, lin.weight, lin.bias))
    b = graph_time(lambda: F.token_linear(x, lin.weight, lin.bias, x_add=pos))
    c = graph_time(lambda: F.token_linear(x, lin.weight, lin.bias, x_add=pos, group_features=min(48, N) if N % 48 == 0 else 0))
    lib = graph_time(lambda: torch.nn.functional.linear(x + pos, lin.weight, lin.bias))
    print("T=%d N=%4d tiles=%2d  store %.1f us | +x_add %.1f | +grouped %.1f | library add+GEMM %.1f" % (T, N, (N + 31) // 32, a, b, c, lib),
          flush=True)
lin = torch.nn.Linear(256, 256).cuda().to(torch.bfloat16)
d = graph_time(lambda: F.token_linear_ln(x, lin, norm, pos))
print("T=%d token_linear_ln (256 -> 256 + residual + LayerNorm) %.1f us" % (T, d))
