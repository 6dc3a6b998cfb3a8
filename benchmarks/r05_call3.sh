#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x > $O/c3_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c3_tests.log
tail -5 $O/c3_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/c3_bench.json 2> $O/c3_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/c3_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step")})
r = d["roofline"]
print({k: r.get(k) for k in ("frac", "avg_launch_us", "per_layer_us", "frac_warm", "valu", "timing_rocprof_us")})
print({k: (v.get("images_per_s"), v.get("ms_per_step"), v.get("served_as"), v.get("error")) for k, v in d.get("configs", {}).items()})
print(d.get("train_step", {}).get("ms_per_step"), d["config"].get("hipgraph_nodes"), d["config"].get("hipgraph_memset_nodes"))
PY
tail -3 $O/c3_bench.err
