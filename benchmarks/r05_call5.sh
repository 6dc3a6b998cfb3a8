#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/c5_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c5_tests.log
tail -12 $O/c5_tests.log
timeout 300 python benchmarks/config4_step.py > $O/c5_config4.json 2> $O/c5_config4.err
cat $O/c5_config4.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_prof -o p -- python benchmarks/config4_step.py --plain --steps 20 > /dev/null 2> $O/c5_prof.err
f=$(find $O/c5_prof -name '*kernel_stats.csv' | head -1)
cp $f $O/c5_config4_kernel_stats.csv
rm -rf $O/c5_prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/c5_config4_kernel_stats.csv')))
steps=25
for r in rows[:14]:
    print(f"{r['Name'][:64]:64s} calls/step {int(r['Calls'])/steps:5.1f} us/step {float(r['TotalDurationNs'])/steps/1e3:8.1f}")
print('total', sum(float(r['TotalDurationNs']) for r in rows)/steps/1e3)
PY
