"""One shape of the fp32-accurate GEMM, a few launches (for rocprofv3 counter passes):
    python benchmarks/gemm_x3_one.py T K N [y|dx|dw] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import linear_x3 as X  # noqa: E402

T, K, N = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "y"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
x = torch.randn(T, K, device="cuda")
w = torch.randn(N, K, device="cuda")
gy = torch.randn(T, N, device="cuda")
for _ in range(reps):
    if mode == "y":
        X.gemm_x3(x, True, w, True, T, N, K)
    elif mode == "dx":
        X.gemm_x3(gy, True, w, False, T, K, N)
    else:
        X.gemm_x3(gy, False, x, False, N, K, T, reduction_splits=X._weight_grad_splits(T, N, K))
torch.cuda.synchronize()
