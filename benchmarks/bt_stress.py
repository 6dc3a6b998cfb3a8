"""Per-launch timing of the LDS backward over many launches: median, max and the number of launches slower than 3x the
median (a rare stall in the scatter waves' LDS-counter barrier or the work list would show here)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import ms_deform_attn as M  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]


def main():
    reps = int(os.environ.get("REPS", "300"))
    out = []
    M.lds_backward, M.lds_backward_min_queries = True, 1
    for nq in [int(a) for a in sys.argv[1:]] or [900, 1300, 4545, 11363]:
        value, shapes, lsi, loc, aw = syn.make_msda_inputs(2, nq, LEVELS, 8, 32, 4, seed=11, spread_px=4.0)
        go = syn.det_randn("gout_ab", (2, nq, 256))
        dev = [t.cuda() for t in (value, shapes, lsi, loc, aw, go)]
        for _ in range(5):
            M.ms_deform_attn_backward(*dev[:5], dev[5], 64)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            M.ms_deform_attn_backward(*dev[:5], dev[5], 64)
            b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        med = t[len(t) // 2]
        out.append({"queries": nq, "launches": reps, "median_us": round(med, 1), "p99_us": round(t[int(0.99 * reps)], 1),
                    "max_us": round(t[-1], 1), "slower_than_3x_median": sum(1 for x in t if x > 3 * med)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
