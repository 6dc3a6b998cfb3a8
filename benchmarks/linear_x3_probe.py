"""x3_linear (as routed by linear_x3.py) against F.linear, forward and backward, on the feed-forward's two layers at the
encoder's token counts.    python benchmarks/linear_x3_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import linear_x3 as X  # noqa: E402
from gemm_x3_bench import time_us  # noqa: E402

for T in (22726, 18180, 13634, 9090, 4544, 2272):
    row = {"T": T}
    for K, N in ((256, 2048), (2048, 256)):
        lin = torch.nn.Linear(K, N).cuda()
        x = torch.randn(2, T // 2, K, device="cuda", requires_grad=True)
        gy = torch.randn(2, T // 2, N, device="cuda")

        def fb(f):
            y = f(x, lin.weight, lin.bias)
            y.backward(gy)
            x.grad = None
            lin.weight.grad = None
            lin.bias.grad = None
        row[f"{K}->{N}"] = (round(time_us(lambda: fb(X.x3_linear), 10), 1), round(time_us(lambda: fb(torch.nn.functional.linear), 10), 1))
    print(row, flush=True)
