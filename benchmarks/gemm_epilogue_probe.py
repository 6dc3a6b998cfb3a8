"""The feed-forward's ReLU inside the x3 products (csrc/gemm_x3.hip epilogues) against the product + torch's elementwise
pass, at the training step's layer sizes.   python benchmarks/gemm_epilogue_probe.py [--out file.json]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from salience_detr_amd import linear_x3 as X   # noqa: E402
from salience_detr_amd import synthetic as syn   # noqa: E402


def time_us(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = "cuda"
    res = []
    for T in (22726, 13634, 9090):
        x = syn.det_randn(f"p.x{T}", (T, 256)).to(dev)
        dy = syn.det_randn(f"p.dy{T}", (T, 256)).to(dev)
        w1 = (syn.det_randn("p.w1", (2048, 256)) * 0.05).to(dev)
        b1 = syn.det_randn("p.b1", (2048,)).to(dev)
        w2 = (syn.det_randn("p.w2", (256, 2048)) * 0.05).to(dev)
        h = torch.empty((T, 2048), device=dev)
        dh = torch.empty((T, 2048), device=dev)
        r = {"tokens": T}
        r["fwd_plain_us"] = time_us(lambda: X.gemm_x3(x, True, w1, True, T, 2048, 256, bias=b1, out=h))
        r["fwd_plain_plus_clamp_us"] = time_us(lambda: X.gemm_x3(x, True, w1, True, T, 2048, 256, bias=b1, out=h).clamp_min_(0.0))
        r["fwd_relu_epilogue_us"] = time_us(lambda: X.gemm_x3(x, True, w1, True, T, 2048, 256, bias=b1, out=h, epilogue=X.EPI_RELU))
        r["bwd_plain_us"] = time_us(lambda: X.gemm_x3(dy, True, w2, False, T, 2048, 256, out=dh))
        r["bwd_plain_plus_threshold_us"] = time_us(
            lambda: torch.ops.aten.threshold_backward(X.gemm_x3(dy, True, w2, False, T, 2048, 256, out=dh), h, 0.0))
        r["bwd_gate_epilogue_us"] = time_us(lambda: X.gemm_x3(dy, True, w2, False, T, 2048, 256, out=dh, epilogue=X.EPI_GATE, gate=h))
        res.append({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
        print(res[-1], flush=True)
    if args.out:
        json.dump({"gemm_epilogue_probe": res}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
