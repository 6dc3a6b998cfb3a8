import sys, json, subprocess, os
sys.path.insert(0, os.getcwd())
import torch, bench
from salience_detr_amd.hot_path import SalienceEncoderHotPath
res = {}
for parts in [(2,1,2,1), (1,1,2,1,1), (1,1,1,1,2), (2,1,1,1,1), (1,1,1,1,1,1), (2,2,2), (3,3)]:
    SalienceEncoderHotPath.value_projection_parts = parts
    sys.argv = ["bench.py", "--plain", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"]
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        try:
            bench.main()
        except SystemExit:
            pass
    line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    res[str(parts)] = d["ms_per_step"]
    print(parts, d["ms_per_step"], flush=True)
