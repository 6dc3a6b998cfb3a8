#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/c4_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c4_tests.log
tail -8 $O/c4_tests.log
for t in 16 0; do
  SDETR_ROW_ORDER_TILE=$t timeout 300 python benchmarks/config4_step.py > $O/c4_config4_tile$t.json 2> $O/c4_config4_tile$t.err
  cat $O/c4_config4_tile$t.json
  SDETR_ROW_ORDER_TILE=$t rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_prof_$t -o p -- python benchmarks/config4_step.py --plain --steps 20 > /dev/null 2> $O/c4_prof_$t.err
  f=$(find $O/c4_prof_$t -name '*kernel_stats.csv' | head -1)
  head -12 $f | cut -d, -f1-5 | cut -c1-160
  cp $f $O/c4_config4_kernel_stats_tile$t.csv
  rm -rf $O/c4_prof_$t
done
