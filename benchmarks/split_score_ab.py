"""Same-box A/B on the graphed headline step: the next layer's class score from the second pass of a split hidden
dimension (filter_ops.SPLIT_PASS_CLASS_SCORE) against the class head as a launch of its own, three pairs in turns.
    python benchmarks/split_score_ab.py
"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from salience_detr_amd import filter_ops as FO  # noqa: E402


def run():
    sys.argv = ["bench.py", "--plain", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        try:
            bench.main()
        except SystemExit:
            pass
    d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
    return d["ms_per_step"]


rows = []
for pair in range(3):
    for on in (False, True):
        FO.SPLIT_PASS_CLASS_SCORE = on
        rows.append({"score_in_second_pass": on, "ms_per_step": run()})
        print(rows[-1], flush=True)
FO.SPLIT_PASS_CLASS_SCORE = True
a = sorted(r["ms_per_step"] for r in rows if not r["score_in_second_pass"])
b = sorted(r["ms_per_step"] for r in rows if r["score_in_second_pass"])
print(json.dumps({"separate_launch_ms_median": a[1], "second_pass_ms_median": b[1], "delta_us": round((a[1] - b[1]) * 1e3, 1)}))
