"""Direct fused MSDA gather vs the LDS-staged (region-tiled) kernel on encoder-like query sets, hipGraph replay.
(Round 1: direct 48 us, tiled 134-203 us incl. bucketing at 11 363 queries -- the staged kernel stays opt-in.)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from salience_detr_amd import ms_deform_attn as M
from salience_detr_amd import synthetic as syn
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
DEV = "cuda:0"
def graph_time(fn, reps=10):
    for _ in range(3): fn()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)
B = 2
for Nq in (11363, 4545):
    for off in (1.5, 3.0):
        tok, ref, proj, shapes, lsi = syn.make_encoder_like_queries(B, Nq, LEVELS, 8, 4, seed=1, offset_px=off)
        hm = M.value_to_head_major(torch.randn(B, 22323, 256, device=DEV), None, 8, torch.bfloat16)
        sh, ls, rf, pj = shapes.to(DEV), lsi.to(DEV), ref.to(DEV), proj.to(torch.bfloat16).to(DEV)
        t_direct = graph_time(lambda: M.msda_fused_forward(hm, sh, ls, rf, pj, 4, 4, out_dtype=torch.bfloat16))
        t_tiled = graph_time(lambda: M.msda_tiled_forward(hm, sh, ls, rf, pj, LEVELS[0], 4, 4, out_dtype=torch.bfloat16))
        print(f"Nq={Nq} off={off}: direct {t_direct:.1f} us, tiled incl. bucketing {t_tiled:.1f} us", flush=True)
