#!/bin/bash
# In-step A/B of the bordered gather's L2 warm-up (sdetr_msda_bordered_forward_ex l2_warmup): rocprofv3 kernel trace of
# `bench.py --plain` per setting, the six MSDA launches' durations per layer.   bash benchmarks/msda_warmup_ab.sh
cd $GRAFT_REPO_ROOT 2>/dev/null; export TMPDIR=/tmp; O=$PWD/gpurun_out
for w in default 0 1; do
  python - $w > $O/_warm_$w.py <<'PY'
import sys
w = sys.argv[1]
print("import sys, runpy, os; sys.path.insert(0, os.getcwd())")
print("from salience_detr_amd.ms_deform_attn import MultiScaleDeformableAttention as M")
print("M.bordered_l2_warmup = %s" % ("None" if w == "default" else w))
print("sys.argv = ['bench.py', '--plain', '--steps', '30']")
print("runpy.run_path('bench.py', run_name='__main__')")
PY
  rocprofv3 --kernel-trace --output-format csv -d $O/_warm_prof_$w -o p -- python $O/_warm_$w.py > /dev/null 2> $O/_warm_$w.err
  python - $(find $O/_warm_prof_$w -name '*kernel_trace.csv' | head -1) $w <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "msda_bordered_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
per = [round(sum(us[k::6]) / len(us[k::6]), 2) for k in range(6)]
print("l2_warmup", sys.argv[2], "per layer", per, "sum", round(sum(per), 1), "launches", len(us))
PY
  rm -rf $O/_warm_prof_$w $O/_warm_$w.py
done
