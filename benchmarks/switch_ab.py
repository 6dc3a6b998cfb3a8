"""Same-box A/B of a module-level switch on the graphed headline step, three pairs in turns (False first).
    python benchmarks/switch_ab.py filter_ops.SPLIT_PASS_CLASS_SCORE      # class score from the split feed-forward's second pass
    python benchmarks/switch_ab.py filter_ops.SELECT_WITH_INPROJECTION    # top-k selection + in-projection in one launch
"""
import contextlib
import importlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

mod_name, attr = sys.argv[1].rsplit(".", 1)
mod = importlib.import_module("salience_detr_amd." + mod_name)
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 3


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
for pair in range(pairs):
    for on in (False, True):
        setattr(mod, attr, on)
        rows.append({attr: on, "ms_per_step": run()})
        print(rows[-1], flush=True)
setattr(mod, attr, True)
a = sorted(r["ms_per_step"] for r in rows if not r[attr])
b = sorted(r["ms_per_step"] for r in rows if r[attr])
print(json.dumps({"switch": sys.argv[1], "off_ms_median": a[len(a) // 2], "on_ms_median": b[len(b) // 2],
                  "delta_us": round((a[len(a) // 2] - b[len(b) // 2]) * 1e3, 1)}))
