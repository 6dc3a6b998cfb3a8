"""A/B of the MFMA 3x3 convolution's forms -- SDETR_CONV_SPLIT = 1: a wave walks its nine taps; 3: three waves per tile,
one kernel row each; SDETR_CONV_LDS = 1 (default): the 64 -> 64 stride-1 blocks with both operands in LDS (the stride-2
"_down" rows never take that form) -- at the benchmark pyramid's four level sizes, plus the neck's gate (context / gate / apply) and the
proposal stage's grid NMS at full size.  Each op is captured 40 times into a hipGraph and replayed; us per call.

    python benchmarks/conv_split_ab.py [--out gpurun_out/conv_split_ab.json]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LEVELS = [(100, 167), (50, 84), (25, 42), (13, 21)]
REPS = 40


def graph_us(fn, reps=REPS, replays=20):
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(3):
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * replays)
        best = us if best is None else min(best, us)
    return round(best, 2)


def child():
    import torch
    from salience_detr_amd import filter_ops as FO
    from salience_detr_amd import synthetic as syn
    dev = "cuda:0"
    rec = {"split": os.environ.get("SDETR_CONV_SPLIT", "default"), "lds_form": os.environ.get("SDETR_CONV_LDS", "default"),
           "conv": {}, "gate": {}}
    w = (syn.det_randn("ab.w", (4, 3, 3, 64, 64)) / 24.0).to(dev)
    wd = (syn.det_randn("ab.wd", (1, 3, 3, 256, 256)) / 48.0).to(dev)
    bias = (0.1 * syn.det_randn("ab.b", (256,))).to(dev)
    packed, packed_d = FO.neck_pack_conv3x3(w), FO.neck_pack_conv3x3(wd)
    mask_w = syn.det_randn("ab.m", (256,)).to(dev) * 0.1
    sq = syn.det_randn("ab.s", (16, 256)).to(dev) * 0.1
    ex = syn.det_randn("ab.e", (256, 16)).to(dev) * 0.1
    for h, ww in LEVELS:
        x = syn.det_randn("ab.x", (2, h * ww, 256)).to(torch.bfloat16).to(dev)
        sc = syn.det_randn("ab.sc", (2, h * ww, 256)).to(torch.bfloat16).to(dev)
        key = f"{h}x{ww}"
        rec["conv"][key] = graph_us(lambda: FO.neck_conv3x3(x, h, ww, w, bias, 1, True, packed=packed))
        if h > 13:
            rec["conv"][key + "_down"] = graph_us(lambda: FO.neck_conv3x3(x, h, ww, wd, bias, 2, True, packed=packed_d))
        rec["gate"][key] = graph_us(lambda: FO.neck_gate_shortcut(x, mask_w, sq, ex, shortcut=sc))
    # grid NMS at the proposal stage's size: the 3600 best of 22223 tokens, clustered like real scores (a smooth field)
    S = sum(h * ww for h, ww in LEVELS)
    g = torch.Generator().manual_seed(3)
    field = torch.cat([torch.nn.functional.avg_pool2d(torch.rand(1, 1, h + 4, ww + 4, generator=g), 5, 1).flatten()
                       for h, ww in LEVELS])
    score = torch.stack([field, field.flip(0)]) + 1e-4 * torch.rand(2, S, generator=g)
    ti = score.topk(3600, 1)[1].to(dev)
    rec["nms_3600_of_22223"] = graph_us(lambda: FO.grid_nms_topk(ti, LEVELS, S, 0.3, 900))
    # decoder self-attention core: 2 images x 8 heads x 900 queries
    qkv = (torch.randn(2, 900, 768, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    rec["attention_2x900"] = graph_us(lambda: FO.attention_heads(qkv[..., :256], qkv[..., 256:512], qkv[..., 512:], 8))
    print("REC " + json.dumps(rec))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/conv_split_ab.json")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child()
    rows = []
    for split, lds in (("1", "0"), ("3", "0"), ("3", "1")):
        env = dict(os.environ, SDETR_CONV_SPLIT=split, SDETR_CONV_LDS=lds)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("REC ")]
        if not line:
            print(out.stdout[-2000:], out.stderr[-4000:])
            raise SystemExit(1)
        rows.append(json.loads(line[0][4:]))
        print(json.dumps(rows[-1]))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
