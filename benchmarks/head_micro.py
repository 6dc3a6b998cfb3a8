"""Time the fused salience head (stage 1 / const + stage 2) per pyramid level of the 800x1333 workload.

    python benchmarks/head_micro.py            (SDETR_HEAD_ROWTILES=1|2 forces the 32/64-token stage-1 variant)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LIB"):   # a scratch build (benchmarks/lib_variant.sh)
    from salience_detr_amd import _hip
    _hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
from salience_detr_amd import filter_ops as F
from salience_detr_amd.salience_filtering import MaskPredictor

DEV = "cuda:0"
LEVELS = [(100, 167), (50, 84), (25, 42), (13, 21)]


def main():
    torch.manual_seed(0)
    B, C = 2, 256
    pred = MaskPredictor(C, C).to(DEV)
    enc, norm = torch.nn.Linear(C, C).to(DEV), torch.nn.LayerNorm(C).to(DEV)
    alpha = torch.tensor([0.2], device=DEV)
    flush = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    for li, (h, w) in enumerate(LEVELS):
        n = h * w
        x = torch.randn(B, n, C, device=DEV)
        coarse = torch.randn(B, 1, (h + 1) // 2, (w + 1) // 2, device=DEV)
        kw = dict(coarse_score=coarse, level_hw=(h, w), alpha=alpha, enc_output=enc, enc_output_norm=norm)
        with torch.no_grad():
            for _ in range(3):
                F.salience_head(x, pred, **kw)
            for cold in (False, True):
                ts = []
                for _ in range(10):
                    if cold:
                        flush.zero_()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    F.salience_head(x, pred, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                ts.sort()
                flops = 2.0 * B * n * (2 * C * C + 128 * 128 + 128 * 64 + 64)
                print("level %d (%dx%d, %d tokens) %s: %.1f us  (%.1f TFLOP/s fp32)" %
                      (li, h, w, n, "cold L2" if cold else "warm   ", ts[len(ts) // 2], flops / ts[len(ts) // 2] / 1e6))


if __name__ == "__main__":
    main()
