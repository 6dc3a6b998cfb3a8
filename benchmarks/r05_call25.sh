#!/bin/bash
# the profile round and the per-kernel profiles of configs[3] / [4] (the artefacts copied into profiles/)
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
bash benchmarks/profile_round.sh r05 > $O/r05_profile_round.log 2>&1
for c in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c25_prof$c -o p -- python benchmarks/config${c}_step.py --plain --steps 20 > /dev/null 2> $O/c25_prof$c.err
  cp $(find $O/c25_prof$c -name '*kernel_stats.csv' | head -1) $O/r05_config${c}_kernel_stats.csv
  python benchmarks/step_timeline.py $(find $O/c25_prof$c -name '*kernel_trace.csv' | head -1) > $O/r05_config${c}_timeline.txt
  rm -rf $O/c25_prof$c
done
timeout 300 python benchmarks/conv_split_ab.py --out $O/r05_conv_split_ab.json | tail -1
python -c "
import json
d=json.load(open('$O/r05_bench.json'))
print('BENCH', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('timing_rocprof_us'))
for k,v in d.get('configs',{}).items(): print('  ',k, v.get('images_per_s'), v.get('ms_per_step'))
print('  train', d.get('train_step',{}).get('ms_per_step'))
"
