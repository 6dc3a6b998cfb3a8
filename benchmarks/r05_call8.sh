#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/c8_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c8_tests.log
tail -4 $O/c8_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python benchmarks/layer_bisect.py --layer 0 --out $O/c8_bisect_l0.json > $O/c8_bisect.log 2>&1; tail -2 $O/c8_bisect.log | cut -c1-600
timeout 600 python bench.py --steps 20 --warmup 5 > $O/c8_bench.json 2> $O/c8_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/c8_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step")})
r = d["roofline"]
print({k: r.get(k) for k in ("frac", "avg_launch_us", "frac_warm", "timing_rocprof_us", "frac_rocprof", "traffic")}, r["valu"]["frac"], r["valu"]["clock_ghz"])
print({k: (v.get("images_per_s"), v.get("error")) for k, v in d.get("configs", {}).items()})
PY
