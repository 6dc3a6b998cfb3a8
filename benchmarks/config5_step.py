"""BASELINE configs[4] at N = 1 (whole SalienceTransformer: neck, encoder, proposals + NMS, six decoder layers at 900
queries; 800x1333 + 800x1066; fp16 = IEEE-half activations since round 5) as a program of its own: ms per step under
hipGraph replay, for per-kernel profiles.

    python benchmarks/config5_step.py [--dtype fp16|bf16|fp32] [--steps 20] [--plain]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402
from salience_detr_amd.salience_transformer import build_salience_transformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--plain", action="store_true")
    ap.add_argument("--no-decoder-head", action="store_true", help="A/B: without the layer head, the GEMM + norm tails and the sine prologue (the state before them)")
    ap.add_argument("--library-linears", action="store_true",
                    help="A/B: the decoder's MLPs and in-projections as library GEMMs (no mlp_rows / rows_linear launches)")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    if args.no_decoder_head or args.library_linears:
        from salience_detr_amd import filter_ops, salience_decoder
        no = lambda *a, **k: False
        filter_ops.decoder_head_applies = salience_decoder.decoder_head_applies = no
        filter_ops.rows_linear_ln_applies = salience_decoder.rows_linear_ln_applies = no
        filter_ops.ref_point_head_applies = salience_decoder.ref_point_head_applies = no
    if args.library_linears:
        filter_ops.rows_linear_applies = filter_ops.mlp_rows_applies = no
        salience_decoder.rows_linear_applies = salience_decoder.mlp_rows_applies = no
    sizes = [(800, 1333), (800, 1066)]
    tr = build_salience_transformer(with_neck=True)
    tr.load_state_dict(syn.det_state_dict(tr.state_dict()))
    tr = tr.eval().to(device)
    if args.dtype != "fp32":
        tr.set_dtype(torch.float16 if args.dtype == "fp16" else torch.bfloat16, torch.float16)
    tr.static_proposals = True
    img_mask, masks = syn.make_masks(sizes)
    canvas = tuple(img_mask.shape[-2:])
    shapes = [tuple(x.shape[-2:]) for x in masks]
    feats = [f.to(device) for f in syn.make_feats(2, shapes, 256, 0)]
    pos = [syn.sine_position_embedding(x, 128).to(device) for x in masks]
    masks = [x.to(device) for x in masks]

    def step():
        with torch.no_grad():
            return tr(feats, masks, pos, image_sizes=sizes, canvas=canvas)
    for _ in range(3):
        step()
    g, _ = bench.capture(step, {})
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(1 if args.plain else 3):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        best = ms if best is None else min(best, ms)
    print(json.dumps({"workload": "BASELINE configs[4] at N=1", "dtype": args.dtype, "library_linears": args.library_linears, "decoder_head": not (args.no_decoder_head or args.library_linears), "ms_per_step": round(best, 4),
                      "images_per_s": round(2e3 / best, 1), "graph_nodes": bench.CAPTURE_INFO.get("graph_nodes")}))


if __name__ == "__main__":
    main()
