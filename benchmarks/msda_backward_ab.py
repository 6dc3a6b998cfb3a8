"""A/B of the two MSDA backward kernels (reference layout, fp32) at the encoder's per-layer query counts, B = 2:
the direct kernel (every contribution an fp32 atomic on global memory) against the LDS-accumulating one
(msda_backward_tiled.hip).  Prints one JSON line; algorithmic bytes per SURVEY.md 8(d):
B*[2*Nv*M*D*4 + 2*Nq*M*L*P*3*4 + Nq*M*D*4]."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LIB"):   # a scratch build (benchmarks/bt_variant.sh)
    from salience_detr_amd import _hip  # noqa: E402
    _hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
from salience_detr_amd import ms_deform_attn as M  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]


def time_us(fn, reps):
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
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--spread", type=float, default=4.0)
    ap.add_argument("--queries", type=int, nargs="*", default=[11363, 9090, 6817, 4545, 2272, 900])
    ap.add_argument("--only-lds", action="store_true")
    a = ap.parse_args()
    rows = []
    for nq in a.queries:
        value, shapes, lsi, loc, aw = syn.make_msda_inputs(a.batch, nq, LEVELS, 8, 32, 4, seed=11, spread_px=a.spread)
        go = syn.det_randn("gout_ab", (a.batch, nq, 256))
        dev = [t.cuda() for t in (value, shapes, lsi, loc, aw, go)]
        nv = value.shape[1]
        alg = a.batch * (2 * nv * 256 * 4 + 2 * nq * 8 * 16 * 3 * 4 + nq * 256 * 4)
        res = {"queries": nq, "algorithmic_bytes": alg}
        for name, flag in ((("lds", True),) if a.only_lds else (("lds", True), ("direct", False))):
            M.lds_backward, M.lds_backward_min_queries = flag, 1
            # the wrapper's zero-fill of grad_value is part of the op as the reference defines it (it accumulates)
            us = time_us(lambda: M.ms_deform_attn_backward(*dev[:5], dev[5], 64), a.reps)
            res[name + "_us"] = round(us, 1)
            res[name + "_frac_of_8TBs"] = round(alg / (us * 1e-6) / 8e12, 4)
        rows.append(res)
    print(json.dumps({"lib": os.path.basename(os.environ.get("LIB", "product")), "batch": a.batch, "spread_px": a.spread, "levels": LEVELS, "layers": rows}))


if __name__ == "__main__":
    main()
