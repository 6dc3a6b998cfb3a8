"""A/B of libsalience_hip.so builds on the fused MSDA gather at the six encoder-layer query counts (+ the decoder's
900): `LIB=<path to .so> python benchmarks/msda_variant_ab.py`.  Events on the launch stream, 100 launches per case."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import _hip

if os.environ.get("LIB"):
    _hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
from salience_detr_amd import ms_deform_attn as M
from salience_detr_amd import synthetic as syn

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
DEV = "cuda:0"
out = {}
hm = M.value_to_head_major(torch.randn(2, 22323, 256, device=DEV), None, 8, torch.bfloat16)
for Nq in (11363, 9090, 6817, 4545, 2272, 900):
    tok, ref, proj, shapes, lsi = syn.make_encoder_like_queries(2, Nq, LEVELS, 8, 4, seed=1, offset_px=1.5)
    sh, ls, rf, pj = shapes.to(DEV), lsi.to(DEV), ref.to(DEV), proj.to(torch.bfloat16).to(DEV)
    fn = lambda: M.msda_fused_forward(hm, sh, ls, rf, pj, 4, 4, out_dtype=torch.bfloat16)
    for _ in range(10):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10.0)
    out[Nq] = round(best, 2)
print(json.dumps({"lib": os.path.basename(_hip.LIB_PATH), "us": out}))
