"""Time the fused feed-forward kernel at the encoder's per-layer token counts (run under rocprofv3 --kernel-trace and
summarise with benchmarks/kernel_times.py, or read the event timings printed here)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as F

DEV = "cuda:0"


def main():
    torch.manual_seed(0)
    lin1 = torch.nn.Linear(256, 2048).to(DEV).to(torch.bfloat16)
    lin2 = torch.nn.Linear(2048, 256).to(DEV).to(torch.bfloat16)
    norm = torch.nn.LayerNorm(256).to(DEV).to(torch.bfloat16)
    from salience_detr_amd import _hip

    def timed(fn):
        with torch.no_grad():
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 20

    for T in (22726, 18180, 13634, 9090, 4544, 1800):
        x = torch.randn(T, 256, device=DEV).to(torch.bfloat16)
        auto = _hip.lib().sdetr_ffn_auto_splits(T, 2048)
        lib_us = timed(lambda: F.fused_layer_norm(x[None], norm, residual=lin2(torch.relu(lin1(x)))[None]))
        line = "T=%6d  library %.1f us | auto=%d" % (T, lib_us, auto)
        for s in sorted({1, 2, 3, 4, 5, 7, 10, 16, auto}):
            us = timed(lambda: F.fused_ffn(x, lin1, lin2, norm, hidden_splits=s))
            line += "  s%d: %.1f" % (s, us)
        print(line + "  (best fused: %.0f TFLOP/s)" % (4.0 * T * 256 * 2048 / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
