"""Rate of the fp32-accurate bf16x3 GEMM against torch's fp32 matmul on the training step's Linear shapes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import linear_x3 as X  # noqa: E402


def time_us(fn, reps=20):
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
    rows = []
    shapes = ((22726, 256, 2048), (22726, 2048, 256), (22726, 256, 384), (22726, 256, 256), (9090, 256, 2048), (44646, 256, 256))
    if len(sys.argv) > 1:   # "T,K,N T,K,N ..."
        shapes = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
    for T, K, N in shapes:
        x = torch.randn(T, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        gy = torch.randn(T, N, device="cuda")
        fl = 2.0 * T * K * N
        r = {"T": T, "K": K, "N": N}
        for name, ours, ref in (
                ("y=xw^T", lambda: X.gemm_x3(x, True, w, True, T, N, K), lambda: x @ w.t()),
                ("dx=dy w", lambda: X.gemm_x3(gy, True, w, False, T, K, N), lambda: gy @ w),
                ("y=xw^T presplit", lambda: X.gemm_x3_presplit_b(x, True, X.presplit(w), T, N, K), lambda: x @ w.t()),
                ("dx presplit", lambda: X.gemm_x3_presplit_b(gy, True, X.presplit(w, transpose=True), T, K, N), lambda: gy @ w),
                ("dw=dy^T x", lambda: X.gemm_x3(gy, False, x, False, N, K, T, reduction_splits=X._weight_grad_splits(T, N, K)),
                 lambda: gy.t() @ x)):
            a, b = time_us(ours), time_us(ref)
            r[name] = {"x3_us": round(a, 1), "x3_tflops": round(fl / a / 1e6, 1), "torch_us": round(b, 1),
                       "torch_tflops": round(fl / b / 1e6, 1)}
        rows.append(r)
    print(json.dumps({"shapes": rows}))


if __name__ == "__main__":
    main()
