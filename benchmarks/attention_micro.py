"""Dense self-attention after the in-projection (32-channel heads): own kernel vs the framework's flash kernel,
hipGraph replay of 10 back-to-back calls.   python benchmarks/attention_micro.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as F


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


for B, N in ((2, 300), (2, 900), (2, 1100)):
    qkv = torch.randn(B, N, 768, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[..., :256], qkv[..., 256:512], qkv[..., 512:]
    own = graph_time(lambda: F.attention_heads(q, k, v, 8))
    h = lambda t: t.view(B, N, 8, 32).transpose(1, 2)
    lib = graph_time(lambda: torch.nn.functional.scaled_dot_product_attention(h(q), h(k), h(v)).transpose(1, 2).reshape(B, N, 256))
    print("B=%d N=%4d: attention_heads %.1f us, framework flash kernel + layout copy %.1f us" % (B, N, own, lib))
