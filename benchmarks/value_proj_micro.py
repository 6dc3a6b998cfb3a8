"""Value projection of the six encoder layers (44 646 tokens x 1536 features, head-major fp16 output) under a hipGraph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as F
x = torch.randn(2, 22323, 256, device='cuda').to(torch.bfloat16)
w = torch.randn(1536, 256, device='cuda').to(torch.bfloat16); b = torch.randn(1536, device='cuda').to(torch.bfloat16)
mask = torch.zeros(2, 22323, dtype=torch.bool, device='cuda')
fn = lambda: F.value_proj_head_major(x, w, b, mask, 8, 6, torch.float16)
for _ in range(3): fn()
s = torch.cuda.Stream()
with torch.cuda.stream(s): fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): fn()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print("value_proj_head_major 44646 tokens x 1536: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
