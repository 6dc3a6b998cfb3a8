"""BASELINE configs[3] at N = 1 (bf16, batch 1, the reference's 5scale pyramid: 89 250 tokens, 45 330 first-layer queries)
as a program of its own: ms per step under hipGraph replay, for A/B runs and per-kernel profiles
(`--row-order-tile 0` = rows in list order, the state before round 5).

    python benchmarks/config4_step.py [--steps 30] [--plain]
    rocprofv3 --kernel-trace --stats ... -- python benchmarks/config4_step.py --plain
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from salience_detr_amd import ms_deform_attn as M  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402
from salience_detr_amd.hot_path import build_hot_path  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--plain", action="store_true", help="no second timing pass: the profile's launches are the step's")
    ap.add_argument("--row-order-tile", type=int, default=16, help="tile edge of the gather's row order; 0 = list order")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    m = build_hot_path(max_num_embedding=500)
    m.encoder.row_order_tile = args.row_order_tile
    m.load_state_dict(syn.det_state_dict(m.state_dict()))
    m = m.to(device).eval()
    m.set_encoder_dtype(torch.bfloat16, torch.float16)
    sizes = [(800, 1333)]
    _, masks = syn.make_masks(sizes, bench.STRESS_LEVELS)
    feats = [f.to(device) for f in syn.make_feats(1, bench.STRESS_LEVELS, 256, seed=0)]
    pos = [syn.sine_position_embedding(x, 128).to(device) for x in masks]
    masks = [x.to(device) for x in masks]
    canvas = syn.pad_to_32(800, 1333)

    def step():
        with torch.no_grad():
            return m(feats, masks, pos, image_sizes=sizes, canvas=canvas)[0]
    for _ in range(3):
        step()
    kernel = M.last_forward_kernel()
    g, _ = bench.capture(step, {})
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(1 if args.plain else 3):
        e0.record()
        for _ in range(args.steps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        best = ms if best is None else min(best, ms)
    print(json.dumps({"workload": "BASELINE configs[3] at N=1", "row_order_tile": args.row_order_tile,
                      "msda_kernel_code": kernel, "tile_order": kernel == M.KERNEL_BORDERED_ORDERED,
                      "ms_per_step": round(best, 4), "images_per_s": round(1e3 / best, 1), "steps": args.steps}))


if __name__ == "__main__":
    main()
