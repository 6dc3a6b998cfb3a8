"""How long would stage 1 of the salience head take for ALL levels' tokens in one launch (VERDICT r5 item 1's hoisted
launch)?  Runs the fused head on one fake level of 22 223 tokens x 2 images (no coarse score: the products, LayerNorms and
GELU are the same work) -- read the stage-1 kernel's time from a rocprofv3 kernel summary of this script:
    rocprofv3 --kernel-trace --stats -d gpurun_out/hp -o p -- python benchmarks/head_all_levels_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as F
from salience_detr_amd.salience_filtering import MaskPredictor

DEV = "cuda:0"
torch.manual_seed(0)
B, C = 2, 256
pred = MaskPredictor(C, C).to(DEV)
enc, norm = torch.nn.Linear(C, C).to(DEV), torch.nn.LayerNorm(C).to(DEV)
for n in (16700, 22223):
    x = torch.randn(B, n, C, device=DEV)
    with torch.no_grad():
        for _ in range(12):
            F.salience_head(x, pred, enc_output=enc, enc_output_norm=norm)
    torch.cuda.synchronize()
