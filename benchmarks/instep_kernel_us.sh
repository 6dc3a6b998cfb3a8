#!/bin/bash
# In-step duration of selected kernels: rocprofv3 --kernel-trace --stats over `bench.py --plain` (hipGraph replay), one run per
# environment setting given as arguments ("VAR=val VAR2=val" each).   bash benchmarks/instep_kernel_us.sh "A=1" "A=2 B=3"
export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  d=gpurun_out/instep_$i
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python bench.py --plain --steps 50 > $d.json 2> $d.err
  python - "$envs" "$(find $d -name '*kernel_stats.csv' | head -1)" $d.json <<'PY'
import csv, json, sys
envs, path, js = sys.argv[1:4]
rows = list(csv.DictReader(open(path)))
steps = int([r for r in rows if "pyramid_flatten" in r["Name"]][0]["Calls"])
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
pick = {}
for r in rows:
    for key in ("msda_bordered", "msda_resident", "row_orders", "fused_attn_proj", "ffn_fused_kernel<true, true>", "topk_hsort", "encoder_prepare", "pyramid_flatten", "finalize_sorted"):
        if key in r["Name"]:
            pick[key] = pick.get(key, 0) + float(r["TotalDurationNs"]) / steps / 1e3
try:
    ms = json.load(open(js))["ms_per_step"]
except Exception:
    ms = None
print(json.dumps({"env": envs, "kernel_sum_us_per_step": round(tot, 1), "profiled_ms_per_step": ms,
                  "us_per_step": {k: round(v, 1) for k, v in pick.items()}}))
PY
  rm -rf $d
done
