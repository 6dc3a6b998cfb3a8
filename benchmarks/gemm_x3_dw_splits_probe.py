"""Weight gradient dw = dy^T x of the narrow Linear layers: time against the number of reduction slices, both tile
generations (linear_x3.pinned_generation).    python benchmarks/gemm_x3_dw_splits_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import linear_x3 as X  # noqa: E402
from gemm_x3_bench import time_us  # noqa: E402

for T, N, K in ((22726, 256, 256), (44646, 256, 256), (22726, 384, 256), (13634, 256, 256), (22726, 128, 256)):
    dy, x = torch.randn(T, N, device="cuda"), torch.randn(T, K, device="cuda")
    out = torch.zeros(N, K, device="cuda")
    row = {"T": T, "N": N, "K": K, "default_splits": X._weight_grad_splits(T, N, K), "torch": round(time_us(lambda: dy.t() @ x), 1)}
    for gen in (1, 2):
        with X.pinned_generation(gen):
            for sp in (8, 16, 32, 64, 88, 128):
                row[f"gen{'A' if gen == 1 else 'B'}_s{sp}"] = round(time_us(lambda: X.gemm_x3(dy, False, x, False, N, K, T, reduction_splits=sp, out=out.zero_())), 1)
    print(row, flush=True)
