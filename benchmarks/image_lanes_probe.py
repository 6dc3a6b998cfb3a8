"""Probe: the encoder of the two images of a batch on two streams.

The filtering stage couples the images of a batch (``score.min()`` over the batch, the per-level budgets), the six encoder
layers do not: every launch of the encoder works image by image (300-row attention per image, rows, MSDA on that image's
maps).  One batch is a strictly dependent chain of ~56 launches most of which fill a fraction of the chip.  This probe
replays the step as THREE hipGraphs -- filtering of the batch, then the encoder of image 0 and of image 1 on two streams --
against the one-graph step, checks that the memory is bit-identical and prints both times.

    python benchmarks/image_lanes_probe.py [--steps 50]
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
from salience_detr_amd.hot_path import build_hot_path  # noqa: E402


def time_ms(fn, steps, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        best = ms if best is None else min(best, ms)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--out", default="gpurun_out/image_lanes_probe.json")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = build_hot_path()
    model.load_state_dict(syn.det_state_dict(model.state_dict()))
    model = model.to(dev).eval()
    model.set_encoder_dtype(torch.bfloat16, torch.float16)
    sizes, canvas, level_shapes, _, (feats, masks, pos) = bench.make_inputs(2, 800, 1333, dev, seed=0)

    def step():
        with torch.no_grad():
            return model(feats, masks, pos, image_sizes=sizes, canvas=canvas)[0]
    for _ in range(3):
        want = step()
    g_all, out_all = bench.capture(step, {})
    t_one = time_ms(g_all.replay, args.steps)

    # ---- the split form
    enc = model.encoder
    real_forward = enc.forward
    kw_store = {}

    def stub(**kw):
        kw_store.clear()
        kw_store.update(kw)
        return kw["query"]
    enc.forward = stub
    try:
        for _ in range(2):
            step()
        g_f, _ = bench.capture(step, {})
    finally:
        enc.forward = real_forward
    kw = dict(kw_store)
    B = kw["query"].shape[0]

    def lane_kwargs(b):
        s = slice(b, b + 1)
        return dict(precomputed_value_maps=None if kw["precomputed_value_maps"] is None else kw["precomputed_value_maps"][:, s],
                    query=kw["query"][s], query_pos=kw["query_pos"][s], query_key_padding_mask=kw["query_key_padding_mask"][s],
                    spatial_shapes=kw["spatial_shapes"], level_start_index=kw["level_start_index"],
                    valid_ratios=kw["valid_ratios"][s], foreground_score=kw["foreground_score"][s],
                    focus_token_nums=kw["focus_token_nums"][s], foreground_inds=[t[s] for t in kw["foreground_inds"]],
                    multi_level_masks=[m[s] for m in kw["multi_level_masks"]], finalize_job=None)
    lanes = []
    for b in range(B):
        st = torch.cuda.Stream()
        kwb = lane_kwargs(b)

        def enc_b(kwb=kwb):
            with torch.no_grad():
                return real_forward(**kwb)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2):
                enc_b()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                out_b = enc_b()
        st.synchronize()
        lanes.append((st, g, out_b))
    main_s = torch.cuda.current_stream()
    ev = torch.cuda.Event()

    def split_step():
        g_f.replay()
        ev.record(main_s)
        for st, g, _ in lanes:
            st.wait_event(ev)
            with torch.cuda.stream(st):
                g.replay()
        for st, _, _ in lanes:
            main_s.wait_stream(st)
    split_step()
    torch.cuda.synchronize()
    got = torch.cat([o for _, _, o in lanes], 0)
    g_all.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(got, out_all))
    maxdiff = float((got.float() - out_all.float()).abs().max())
    t_split = time_ms(split_step, args.steps)
    # the filtering graph alone and one encoder lane alone, for the breakdown
    t_f = time_ms(g_f.replay, args.steps)

    def one_lane():
        st, g, _ = lanes[0]
        with torch.cuda.stream(st):
            g.replay()
        main_s.wait_stream(st)
    t_lane = time_ms(one_lane, args.steps)
    rec = {"one_graph_ms": round(t_one, 4), "split_ms": round(t_split, 4), "filtering_graph_ms": round(t_f, 4),
           "one_encoder_lane_ms": round(t_lane, 4), "bit_identical": same, "max_abs_diff": maxdiff,
           "images_per_s_one_graph": round(2e3 / t_one, 1), "images_per_s_split": round(2e3 / t_split, 1)}
    print(json.dumps(rec))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
