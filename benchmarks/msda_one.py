"""Run only the fused MSDA kernels (direct + LDS-staged) a few times at the benchmark shape, for rocprofv3 --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import ms_deform_attn as M
from salience_detr_amd import synthetic as syn

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
DEV = "cuda:0"
B = int(os.environ.get("B", 2))
Nq = int(os.environ.get("NQ", 11363))
reps = int(os.environ.get("REPS", 5))
tok, ref, proj, shapes, lsi = syn.make_encoder_like_queries(B, Nq, LEVELS, 8, 4, seed=1, offset_px=1.5)
VDT = torch.float16 if os.environ.get("VALUE", "f16") == "f16" else torch.bfloat16
hm = M.value_to_head_major(torch.randn(B, 22323, 256, device=DEV), None, 8, VDT)
sh, ls, rf, pj = shapes.to(DEV), lsi.to(DEV), ref.to(DEV), proj.to(torch.bfloat16).to(DEV)
for _ in range(reps):
    M.msda_fused_forward(hm, sh, ls, rf, pj, 4, 4, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
