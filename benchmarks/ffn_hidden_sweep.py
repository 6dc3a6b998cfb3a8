"""Fused feed-forward kernel time against the hidden size at a fixed token count (separates the per-chunk loop cost
from the prologue + epilogue): python benchmarks/ffn_hidden_sweep.py [tokens]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as F

T = int(sys.argv[1]) if len(sys.argv) > 1 else 22726
torch.manual_seed(0)
norm = torch.nn.LayerNorm(256).cuda().to(torch.bfloat16)
x = torch.randn(T, 256, device="cuda").to(torch.bfloat16)
for hidden in (32, 64, 256, 512, 1024, 2048, 4096):
    lin1 = torch.nn.Linear(256, hidden).cuda().to(torch.bfloat16)
    lin2 = torch.nn.Linear(hidden, 256).cuda().to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(3):
            F.fused_ffn(x, lin1, lin2, norm, hidden_splits=1)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            F.fused_ffn(x, lin1, lin2, norm, hidden_splits=1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(10):
                F.fused_ffn(x, lin1, lin2, norm, hidden_splits=1)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    print("T=%d hidden=%5d chunks=%3d  %.1f us" % (T, hidden, hidden // 32, e0.elapsed_time(e1) * 1e3 / 50), flush=True)
