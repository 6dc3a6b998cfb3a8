"""Stage 1 of the salience head per level, with enc_output inside the launch and starting from its output (the split the
round-6 hoist experiment needed: benchmarks/experiments/README.md).  Kernel times from the profiler."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
if os.environ.get("LIB"):
    from salience_detr_amd import _hip
    _hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
from salience_detr_amd import filter_ops as F
from salience_detr_amd.salience_filtering import MaskPredictor
DEV = "cuda:0"
torch.manual_seed(0)
B, C = 2, 256
pred = MaskPredictor(C, C).to(DEV)
enc, norm = torch.nn.Linear(C, C).to(DEV), torch.nn.LayerNorm(C).to(DEV)
alpha = torch.tensor([0.2], device=DEV)
import torch.cuda as tc
def time_kernels(fn, reps=20):
    for _ in range(3): fn()
    tc.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps): fn()
        tc.synchronize()
    out = {}
    for e in prof.key_averages():
        if "stage1" in e.key or "stage2" in e.key or "const" in e.key:
            out[e.key.split("(")[0][-40:]] = round(e.device_time_total / e.count, 1)
    return out
for (h, w) in [(100, 167), (50, 84), (13, 21)]:
    n = h * w
    x = torch.randn(B, n, C, device=DEV)
    coarse = torch.randn(B, 1, (h + 1) // 2, (w + 1) // 2, device=DEV)
    with torch.no_grad():
        full = time_kernels(lambda: F.salience_head(x, pred, coarse_score=coarse, level_hw=(h, w), alpha=alpha, enc_output=enc, enc_output_norm=norm))
        half = time_kernels(lambda: F.salience_head(x, pred, coarse_score=coarse, level_hw=(h, w), alpha=alpha))
    print(n, "with enc_output:", full)
    print(n, "from memory    :", half)
