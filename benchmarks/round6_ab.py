"""Same-box A/B of what round 6 changed on the headline step (the pool's boxes differ by ~4 %: two trees can only be
compared on one box).  "round 5 form" = the finest level's top-k as the one-workgroup histogram sort, foreground_score
materialised by its own launch, the gather's L2 warm-up at every layer size; "round 6 form" = the tree as it is.  The two
forms are timed in turns (hipGraph replay, `bench.py --plain`), three pairs.     python benchmarks/round6_ab.py
"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from salience_detr_amd import filter_ops, hot_path, salience_filtering as SF  # noqa: E402
from salience_detr_amd.ms_deform_attn import MultiScaleDeformableAttention as MSDA  # noqa: E402

real_filtering = SF.salience_filtering
real_min_row = filter_ops._SLICED_MIN_ROW


def set_form(old: bool):
    filter_ops._SLICED_MIN_ROW = 10 ** 9 if old else real_min_row
    MSDA.bordered_l2_warmup = 1 if old else None
    if old:
        def eager_fill(*a, **kw):
            kw["lazy_foreground"] = False
            return real_filtering(*a, **kw)
        hot_path.salience_filtering = eager_fill
    else:
        hot_path.salience_filtering = real_filtering


def run():
    sys.argv = ["bench.py", "--plain", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        try:
            bench.main()
        except SystemExit:
            pass
    d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
    return d["ms_per_step"], (d.get("config") or {}).get("hipgraph_nodes")


rows = []
for pair in range(3):
    for old in (True, False):
        set_form(old)
        ms, nodes = run()
        rows.append({"form": "round 5" if old else "round 6", "ms_per_step": ms, "graph_nodes": nodes})
        print(rows[-1], flush=True)
old = sorted(r["ms_per_step"] for r in rows if r["form"] == "round 5")
new = sorted(r["ms_per_step"] for r in rows if r["form"] == "round 6")
print(json.dumps({"round5_form_ms_median": old[1], "round6_form_ms_median": new[1],
                  "delta_us": round((old[1] - new[1]) * 1e3, 1), "runs": rows}))
