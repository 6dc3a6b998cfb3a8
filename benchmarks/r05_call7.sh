#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for dt in fp16 bf16; do
timeout 300 python benchmarks/config5_step.py --dtype $dt > $O/c7_config5_$dt.json 2> $O/c7_config5_$dt.err
cat $O/c7_config5_$dt.json
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c7_prof -o p -- python benchmarks/config5_step.py --plain --steps 20 > /dev/null 2> $O/c7_prof.err
f=$(find $O/c7_prof -name '*kernel_stats.csv' | head -1)
cp $f $O/c7_config5_kernel_stats.csv
python benchmarks/step_timeline.py $(find $O/c7_prof -name '*kernel_trace.csv' | head -1) > $O/c7_config5_timeline.txt
rm -rf $O/c7_prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/c7_config5_kernel_stats.csv')))
steps=25
for r in rows[:30]:
    print(f"{r['Name'][:84]:84s} calls/step {int(r['Calls'])/steps:5.1f} us/step {float(r['TotalDurationNs'])/steps/1e3:8.1f}")
print('total', sum(float(r['TotalDurationNs']) for r in rows)/steps/1e3)
PY
timeout 300 python benchmarks/config4_step.py
