"""Same-box A/B of the FFN's hidden-dimension split choice on the headline step: `sdetr_ffn_auto_splits` as built against
the same call forced to 1 for token counts in [LO, HI) -- with 1 the layer end takes the NEXT form (class head + advance
in the FFN's epilogue: no reduce launch, no class-head launch).     LO=12000 HI=20000 python benchmarks/ffn_split_ab.py
MODE=join compares instead the two forms of a SPLIT hidden dimension: joined inside the launch (filter_ops.FFN_JOIN, "forced" =
True in the output) against the reduce + LayerNorm pass and the class head in launches of their own.
"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from salience_detr_amd import _hip, filter_ops  # noqa: E402

LO, HI = int(os.environ.get("LO", "12000")), int(os.environ.get("HI", "20000"))
library = _hip.lib(torch.bfloat16)
real = library.sdetr_ffn_auto_splits
seen = {}


def forced(tokens, hidden):
    s = real(tokens, hidden)
    seen[(tokens, hidden)] = s
    return 1 if LO <= tokens < HI else s


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
    for force in (False, True):
        if os.environ.get("MODE") == "join":
            filter_ops.FFN_JOIN = force
        else:
            library.sdetr_ffn_auto_splits = forced if force else real
        ms, nodes = run()
        rows.append({"forced": force, "ms_per_step": ms, "graph_nodes": nodes})
        print(rows[-1], flush=True)
a = sorted(r["ms_per_step"] for r in rows if not r["forced"])
b = sorted(r["ms_per_step"] for r in rows if r["forced"])
filter_ops.FFN_JOIN = True
print(json.dumps({"mode": os.environ.get("MODE", "force one piece"), "range": [LO, HI], "auto_choices": {f"{k[0]}x{k[1]}": v for k, v in sorted(seen.items())},
                  "auto_ms_median": a[1], "forced_ms_median": b[1], "delta_us": round((a[1] - b[1]) * 1e3, 1)}))
