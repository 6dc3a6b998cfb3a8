#!/bin/bash
# final artefacts of the round: full GPU suite + smoke, the profile round, the per-kernel profiles of configs[3] / [4]
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/c24_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c24_tests.log
tail -4 $O/c24_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash benchmarks/profile_round.sh r05 > $O/r05_profile_round.log 2>&1
for c in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c24_prof$c -o p -- python benchmarks/config${c}_step.py --plain --steps 20 > /dev/null 2> $O/c24_prof$c.err
  cp $(find $O/c24_prof$c -name '*kernel_stats.csv' | head -1) $O/r05_config${c}_kernel_stats.csv
  python benchmarks/step_timeline.py $(find $O/c24_prof$c -name '*kernel_trace.csv' | head -1) > $O/r05_config${c}_timeline.txt
  rm -rf $O/c24_prof$c
done
timeout 300 python benchmarks/config5_step.py --dtype fp16 --library-linears
timeout 300 python benchmarks/config5_step.py --dtype fp16 --no-decoder-head
timeout 300 python benchmarks/config5_step.py --dtype fp16
timeout 300 python benchmarks/config5_step.py --dtype fp32
timeout 300 python benchmarks/config4_step.py
timeout 300 python benchmarks/conv_split_ab.py --out $O/r05_conv_split_ab.json | tail -1
python -c "
import json
d=json.load(open('$O/r05_bench.json'))
print('BENCH', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('timing_rocprof_us'))
for k,v in d.get('configs',{}).items(): print('  ',k, v.get('images_per_s'), v.get('ms_per_step'))
print('  train', d.get('train_step',{}).get('ms_per_step'))
"
