"""Same-box A/B of the hoisted salience head (salience_filtering.HOIST_HEAD) on the graphed headline step, three pairs in
turns, and the score / index agreement of the two forms on the bench inputs.     python benchmarks/hoist_ab.py
"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from salience_detr_amd import salience_filtering as SF  # noqa: E402


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
    for hoist in (False, True):
        SF.HOIST_HEAD = hoist
        if os.environ.get("CARRY_VALUE"):
            SF.HOIST_CARRIES_VALUE = True
        if os.environ.get("FIN_LEVEL"):
            v = os.environ["FIN_LEVEL"]
            SF.FINALIZE_LEVEL = None if v == "hoist" else (v if v == "merge" else int(v))
        if os.environ.get("RANK_ON_E0"):
            SF.RANK_ON_MERGE = False
        if os.environ.get("CONST_LAUNCH"):
            from salience_detr_amd import filter_ops as FO
            FO.CONST_IN_BLOCK = False
        if os.environ.get("PARTS"):
            from salience_detr_amd.hot_path import SalienceEncoderHotPath as SalienceHotPath
            SalienceHotPath.value_projection_parts_hoisted = tuple(int(v) for v in os.environ["PARTS"].split(","))
        ms = run()
        rows.append({"hoisted": hoist, "ms_per_step": ms})
        print(rows[-1], flush=True)
SF.HOIST_HEAD = True
a = sorted(r["ms_per_step"] for r in rows if not r["hoisted"])
b = sorted(r["ms_per_step"] for r in rows if r["hoisted"])
print(json.dumps({"per_level_ms_median": a[1], "hoisted_ms_median": b[1], "delta_us": round((a[1] - b[1]) * 1e3, 1)}))
