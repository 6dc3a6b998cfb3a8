"""Probe: fp32-accurate Linear products as ONE library bf16 GEMM over a 6x longer reduction (operands pre-split into three
bf16 planes and concatenated along k: [a0 a0 a1 a1 a0 a2] . [b0 b1 b0 b1 b2 b0]) with fp32 output -- against
csrc/gemm_x3.hip and the library's fp32 GEMM.  Prints one JSON line per shape."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd.linear_x3 import gemm_x3  # noqa: E402

DEV = "cuda:0"


def split3(x):
    h0 = (x.view(torch.int32) & -65536).view(torch.float32)
    r1 = x - h0
    h1 = (r1.view(torch.int32) & -65536).view(torch.float32)
    r2 = r1 - h1
    return h0.bfloat16(), h1.bfloat16(), r2.bfloat16()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    print("torch", torch.__version__)
    for T, K, N in [(22726, 256, 2048), (22726, 2048, 256), (22726, 256, 256), (9090, 256, 2048), (44646, 256, 256)]:
        x = torch.randn(T, K, device=DEV)
        w = torch.randn(N, K, device=DEV) * 0.05
        ref = (x.double() @ w.double().t())
        row = {"T": T, "K": K, "N": N}
        y32 = x @ w.t()
        row["fp32_us"] = round(timeit(lambda: x @ w.t()), 1)
        row["fp32_err"] = float((y32.double() - ref).abs().max() / ref.abs().max())
        y3 = gemm_x3(x, True, w, True, T, N, K)
        row["x3_us"] = round(timeit(lambda: gemm_x3(x, True, w, True, T, N, K)), 1)
        row["x3_err"] = float((y3.double() - ref).abs().max() / ref.abs().max())
        a0, a1, a2 = split3(x)
        b0, b1, b2 = split3(w)
        A = torch.cat([a0, a0, a1, a1, a0, a2], 1).contiguous()
        Bm = torch.cat([b0, b1, b0, b1, b2, b0], 1).contiguous()
        try:
            yk = torch.mm(A, Bm.t(), out_dtype=torch.float32)
            row["kconcat_us"] = round(timeit(lambda: torch.mm(A, Bm.t(), out_dtype=torch.float32)), 1)
            row["kconcat_err"] = float((yk.double() - ref).abs().max() / ref.abs().max())
        except Exception as e:  # noqa: BLE001
            row["kconcat_error"] = repr(e)[:200]
        row["split_cat_us"] = round(timeit(lambda: torch.cat([*(s := split3(x))[:1] * 2, s[1], s[1], s[0], s[2]], 1)), 1)
        # plain bf16 GEMM of the same flops (the library's rate on this shape)
        yb = A @ Bm.t()
        row["bf16_out_us"] = round(timeit(lambda: A @ Bm.t()), 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
