"""One fused feed-forward launch configuration repeated (for rocprofv3 --pmc / --kernel-trace runs).

    python benchmarks/ffn_one.py [tokens] [hidden_splits] [repeats]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as F

T = int(sys.argv[1]) if len(sys.argv) > 1 else 22726
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.manual_seed(0)
lin1 = torch.nn.Linear(256, 2048).cuda().to(torch.bfloat16)
lin2 = torch.nn.Linear(2048, 256).cuda().to(torch.bfloat16)
norm = torch.nn.LayerNorm(256).cuda().to(torch.bfloat16)
x = torch.randn(T, 256, device="cuda").to(torch.bfloat16)
with torch.no_grad():
    for _ in range(N):
        F.fused_ffn(x, lin1, lin2, norm, hidden_splits=S)
torch.cuda.synchronize()
