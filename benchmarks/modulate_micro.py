"""The modulation step of the hoisted salience head alone, per level of the 800x1333 workload (run under
`rocprofv3 --kernel-trace` and read fused_modulate_rank_kernel's durations, or take the event times printed here)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as F
from salience_detr_amd.salience_filtering import MaskPredictor

DEV = "cuda:0"
torch.manual_seed(0)
B, C = 2, 256
pred = MaskPredictor(C, C).to(DEV)
for (h, w) in [(100, 167), (50, 84), (25, 42), (13, 21)]:
    n = h * w
    x = torch.randn(B, n, C, device=DEV)
    coarse = torch.randn(B, 1, (h + 1) // 2, (w + 1) // 2, device=DEV)
    alpha = torch.tensor([0.3], device=DEV)
    with torch.no_grad():
        hh = F.salience_head_hoist(x, pred)
        lib = F._hip.lib()
        z = torch.empty((B, n, 128), device=DEV)
        part = torch.empty((B, (n + 31) // 32, 128), device=DEV)
        def go():
            F._hip.check(lib.sdetr_salience_head_modulate(
                F._hip.stream_ptr(), hh.g.data_ptr(), hh.g.stride(0), hh.sigma.data_ptr(), hh.sigma.stride(0), B, n, None,
                coarse.data_ptr(), coarse.shape[2], coarse.shape[3], h, w, alpha.data_ptr(), 1e-5, hh.c0.data_ptr(),
                z.data_ptr(), part.data_ptr(), None, None), "modulate")
        for _ in range(5):
            go()
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); go(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print("level %dx%d (%d tokens x %d): modulate %.1f us (event-timed, launch included)" % (h, w, n, B, ts[len(ts) // 2]))
