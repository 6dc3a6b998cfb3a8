#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_neck_gpu.py tests/test_transformer_gpu.py -q > $O/c9_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c9_tests.log
tail -3 $O/c9_tests.log
timeout 300 python benchmarks/config5_step.py --dtype fp16
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c9_prof -o p -- python benchmarks/config5_step.py --plain --steps 20 > /dev/null 2> $O/c9_prof.err
f=$(find $O/c9_prof -name '*kernel_stats.csv' | head -1)
cp $f $O/c9_config5_kernel_stats.csv
python benchmarks/step_timeline.py $(find $O/c9_prof -name '*kernel_trace.csv' | head -1) > $O/c9_config5_timeline.txt
rm -rf $O/c9_prof
grep "se_context\|se_gate\|se_apply" $O/c9_config5_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
bash benchmarks/profile_round.sh r05 > $O/r05_profile_round.log 2>&1
tail -3 $O/r05_profile_round.log
