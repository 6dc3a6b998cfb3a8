"""A/B of the round-3 resident MSDA kernel against the round-4 bordered-map kernel, without and with a spatial row order,
at the benchmark's six encoder layer sizes: same operands, `reps` launches in a captured hipGraph between two events.

    python benchmarks/msda_bordered_ab.py [--batch 2] [--reps 30] [--tiles 8,16,32] [--out gpurun_out/msda_bordered_ab.json]
"""
import argparse
import json
import os
import sys

# the ablated kernels and the phase stamps live in the benchmark build of the library only (benchmarks/libsalience_hip_ablate.so,
# `python salience_detr_amd/csrc/build.py --ablations`): bound below, before the first operator call
USE_ABLATION_BUILD = any(a.startswith(("--ablate", "--stamps")) for a in sys.argv[1:])

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import ms_deform_attn as M  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402
from benchmarks.msda_resident_ab import slab_of, timeit  # noqa: E402
from salience_detr_amd import _hip  # noqa: E402

if USE_ABLATION_BUILD:
    _hip.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsalience_hip_ablate.so")

DEV = "cuda:0"
HEADS, L, P = 8, 4, 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--nq", default="11363,9090,6817,4545,2272")
    ap.add_argument("--tiles", default="8,16,32")
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--levels", default="4scale", choices=["4scale", "5scale"])
    ap.add_argument("--ablate", default="", help="comma list of SDETR_MSDA_ABLATE masks timed with the tile-16 order "
                    "(1 no fine-level loads, 2 no LDS map reads, 4 / 8 no products of the fine / resident levels, 16 one "
                    "record for everybody); results are wrong by construction")
    ap.add_argument("--minimal", action="store_true", help="only: round-3 resident kernel, bordered kernel in list order, "
                    "bordered kernel in the first --tiles order (what the counter passes wrap: one configuration per kernel name)")
    ap.add_argument("--stamps", action="store_true", help="phase stamps of the workgroups (wall clock, one eager launch)")
    ap.add_argument("--out", default="gpurun_out/msda_bordered_ab.json")
    args = ap.parse_args()
    B = args.batch
    levels = [(100, 168), (50, 84), (25, 42), (13, 21)] if args.levels == "4scale" else [(200, 336), (100, 168), (50, 84), (25, 42)]
    Nv = sum(h * w for h, w in levels)
    hm = M.value_to_head_major(torch.randn(B, Nv, 256, device=DEV), None, HEADS, torch.float16)
    hb = M.to_bordered(hm, levels)
    rows = []
    for nq in [int(x) for x in args.nq.split(",")]:
        tok, ref, proj, shapes, lsi = syn.make_encoder_like_queries(B, nq, levels, HEADS, P, seed=1, offset_px=1.0)
        proj[..., :HEADS * L * P * 2] += syn._ring_bias(HEADS, L, P)
        slab = slab_of(proj.to(torch.bfloat16)).to(DEV)
        rf, tk = ref.to(DEV), tok.to(DEV)
        alg = B * (Nv * 256 * 2 + nq * HEADS * L * P * 3 * 2 + nq * L * 2 * 4 + nq * 256 * 2)
        row = {"batch": B, "nq": nq, "algorithmic_MB": round(alg / 1e6, 2)}

        def record(name, fn, base=None):
            out = fn()
            t = timeit(fn, args.reps)
            row[name + "_us"] = round(t, 2)
            row[name + "_frac"] = round(alg / t / 1e6 / 8.0, 4)
            if base is not None:
                row[name + "_maxdiff"] = float((out.float() - base.float()).abs().max())
            return out

        base = record("resident_r3", lambda: M.msda_resident_forward(hm, levels, rf, slab, out_dtype=torch.bfloat16))
        for ch in [int(c) for c in args.chunks.split(",")]:
            sfx = "" if ch == 0 else f"_c{ch}"
            record("bordered" + sfx, lambda: M.msda_bordered_forward(hb, levels, rf, slab, out_dtype=torch.bfloat16, chunks=ch), base)
            raster = tk.argsort(1).to(torch.int32)
            if not args.minimal:
                record("bordered_raster" + sfx, lambda: M.msda_bordered_forward(hb, levels, rf, slab, row_order=raster,
                                                                                out_dtype=torch.bfloat16, chunks=ch), base)
            for tile in [int(t) for t in args.tiles.split(",")][:1 if args.minimal else None]:
                order = M.spatial_row_order(tk, levels, tile)
                record(f"bordered_tile{tile}" + sfx, lambda: M.msda_bordered_forward(hb, levels, rf, slab, row_order=order,
                                                                                      out_dtype=torch.bfloat16, chunks=ch), base)
        order16 = M.spatial_row_order(tk, levels, 16)
        if not args.minimal:
            record("tile16_no_l2_warmup", lambda: M.msda_bordered_forward(hb, levels, rf, slab, row_order=order16,
                                                                          out_dtype=torch.bfloat16, l2_warmup=0), base)
            record("tile16_exact_fp32_sums", lambda: M.msda_bordered_forward(hb, levels, rf, slab, row_order=order16,
                                                                             out_dtype=torch.bfloat16, accumulate=M.ACC_EXACT), base)
        if args.ablate:
            order = M.spatial_row_order(tk, levels, 16)
            for abl in args.ablate.split(","):
                os.environ["SDETR_MSDA_ABLATE"] = abl
                fn = lambda: M.msda_bordered_forward(hb, levels, rf, slab, row_order=order, out_dtype=torch.bfloat16)
                fn()
                row[f"ablate{abl}_us"] = round(timeit(fn, args.reps), 2)
                os.environ.pop("SDETR_MSDA_ABLATE")
        if args.stamps:
            import ctypes
            from salience_detr_amd import _hip
            order = M.spatial_row_order(tk, levels, 16)
            st = torch.zeros(256 * 8 + 64, dtype=torch.int64, device=DEV)
            fn_ = _hip.lib().sdetr_msda_debug_stamps
            fn_.restype, fn_.argtypes = None, [ctypes.c_void_p]
            fn_(st.data_ptr())
            os.environ["SDETR_MSDA_ABLATE"] = "32"
            for _ in range(3):
                M.msda_bordered_forward(hb, levels, rf, slab, row_order=order, out_dtype=torch.bfloat16)
            torch.cuda.synchronize()
            os.environ.pop("SDETR_MSDA_ABLATE")
            w9 = st[2048:].view(16, 4).cpu().double() * 0.01
            t = st[:2048].view(256, 8).cpu().double() * 0.01     # 100 MHz -> us
            t0 = t[:, 0].min()
            row["wg9_waves"] = {"entry": [round(float(v - w9[:, 0].min()), 2) for v in w9[:, 0]],
                                "at_barrier": [round(float(v - w9[:, 0].min()), 2) for v in w9[:, 1]],
                                "past_barrier": [round(float(v - w9[:, 0].min()), 2) for v in w9[:, 2]]}
            row["stamps_us"] = {"entry_spread": round(float(t[:, 0].max() - t0), 2),
                                "entry_to_index_mean": round(float((t[:, 4] - t[:, 0]).mean()), 2),
                                "index_to_inputs_mean": round(float((t[:, 5] - t[:, 4]).mean()), 2),
                                "inputs_to_first_loads_issued_mean": round(float((t[:, 1] - t[:, 5]).mean()), 2),
                                "entry_to_first_loads_issued_mean": round(float((t[:, 1] - t[:, 0]).mean()), 2),
                                "maps_wait_mean": round(float((t[:, 2] - t[:, 1]).mean()), 2),
                                "maps_landed_after_start_mean": round(float((t[:, 2] - t0).mean()), 2),
                                "loop_mean": round(float((t[:, 3] - t[:, 2]).mean()), 2),
                                "end_first": round(float(t[:, 3].min() - t0), 2),
                                "end_last": round(float(t[:, 3].max() - t0), 2),
                                # workgroup = (head = block % 8, image, chunk): end time per head (mean over its 32
                                # workgroups) and per chunk position (mean over heads and images)
                                "end_by_head": [round(float((t[h::8, 3] - t0).mean()), 2) for h in range(8)],
                                "end_by_chunk": [round(float((t.view(2, 16, 8, 8)[:, c, :, 3] - t0).mean()), 2) for c in range(16)],
                                "end_by_image": [round(float((t.view(2, 16, 8, 8)[i, :, :, 3] - t0).mean()), 2) for i in range(2)],
                                "landed_by_head": [round(float((t[h::8, 2] - t0).mean()), 2) for h in range(8)]}
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
