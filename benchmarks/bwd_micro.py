"""MSDA backward (col2im) timings at the encoder's query counts (stream events; includes the grad_value zero-fill)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import ms_deform_attn as M
from salience_detr_amd import synthetic as syn
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
for Nq in (11363, 4545, 2272):
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(2, Nq, LEVELS, 8, 32, 4, seed=0, spread_px=3.0)
    args = [t.cuda() for t in (value, shapes, lsi, loc, aw)] + [torch.randn(2, Nq, 256, device="cuda")]
    for name, kw in (("atomic", {}),):
        for _ in range(2):
            M.ms_deform_attn_backward(*args, 64, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            M.ms_deform_attn_backward(*args, 64, **kw)
        e1.record()
        torch.cuda.synchronize()
        print(Nq, name, round(e0.elapsed_time(e1) * 1e3 / 5, 1), "us (incl. zero-fill + bucketing)")
