"""A/B of the fused MSDA forward kernels at the benchmark's six encoder layer sizes (and the decoder's 900 queries):
direct gather (msda_gather_l4p4_kernel) vs coarse-levels-in-LDS (msda_resident_kernel), same operands, back-to-back
launches between two stream events.  Prints a table and writes gpurun_out/msda_ab.json.

    python benchmarks/msda_resident_ab.py [--batch 2] [--reps 30] [--chunks 0,8,16,32]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import ms_deform_attn as M  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
DEV = "cuda:0"
HEADS, L, P = 8, 4, 4


def slab_of(proj):
    B, Nq, _ = proj.shape
    off = proj[..., :HEADS * L * P * 2].view(B, Nq, HEADS, L * P * 2)
    lgt = proj[..., HEADS * L * P * 2:].view(B, Nq, HEADS, L * P)
    return torch.cat([off, lgt], -1).permute(0, 2, 1, 3).contiguous()


def timeit(fn, reps):
    """Mean device time of one launch: `reps` launches captured in a hipGraph (no host launch gaps -- the python
    wrapper costs more host time than the small layers' kernels take), replayed between two events."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--nq", default="11363,9090,6817,4545,2272,900")
    ap.add_argument("--sorted", action="store_true", help="queries in spatial (token) order instead of random order")
    ap.add_argument("--levels", default="4scale", choices=["4scale", "5scale"],
                    help="5scale: the reference's strides 4-32 pyramid (level 3 alone resident in LDS)")
    ap.add_argument("--out", default="gpurun_out/msda_ab.json")
    args = ap.parse_args()
    B = args.batch
    global LEVELS
    if args.levels == "5scale":
        LEVELS = [(200, 336), (100, 168), (50, 84), (25, 42)]
    Nv = sum(h * w for h, w in LEVELS)
    hm = M.value_to_head_major(torch.randn(B, Nv, 256, device=DEV), None, HEADS, torch.float16)
    rows = []
    for nq in [int(x) for x in args.nq.split(",")]:
        tok, ref, proj, shapes, lsi = syn.make_encoder_like_queries(B, nq, LEVELS, HEADS, P, seed=1, offset_px=1.0)
        if args.sorted:   # queries in token (= spatial, row-major per level) order instead of random order
            perm = tok.argsort(1)
            ref = torch.gather(ref, 1, perm[:, :, None, None].expand_as(ref))
            proj = torch.gather(proj, 1, perm[:, :, None].expand_as(proj))
        proj[..., :HEADS * L * P * 2] += syn._ring_bias(HEADS, L, P)     # the benchmark's offsets: ring + noise
        slab = slab_of(proj.to(torch.bfloat16)).to(DEV)
        sh, ls, rf = shapes.to(DEV), lsi.to(DEV), ref.to(DEV)
        direct = lambda: M.msda_fused_forward(hm, sh, ls, rf, slab, L, P, out_dtype=torch.bfloat16, proj_head_major=True)
        a = direct()
        t_direct = timeit(direct, args.reps)
        alg = B * (Nv * 256 * 2 + nq * HEADS * L * P * 3 * 2 + nq * L * 2 * 4 + nq * 256 * 2)
        row = {"batch": B, "nq": nq, "algorithmic_MB": round(alg / 1e6, 2), "direct_us": round(t_direct, 2),
               "direct_frac": round(alg / t_direct / 1e6 / 8.0, 4)}
        for ch in [int(c) for c in args.chunks.split(",")]:
            res = lambda: M.msda_resident_forward(hm, LEVELS, rf, slab, out_dtype=torch.bfloat16, chunks=ch)
            b = res()
            t = timeit(res, args.reps)
            row[f"resident_c{ch}_us"] = round(t, 2)
            row[f"resident_c{ch}_frac"] = round(alg / t / 1e6 / 8.0, 4)
            row[f"resident_c{ch}_maxdiff"] = float((a.float() - b.float()).abs().max())
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
