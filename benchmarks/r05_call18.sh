#!/bin/bash
# final artefacts of the round: the profile round, then the per-kernel profiles of configs[3] and configs[4]
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
bash benchmarks/profile_round.sh r05 > $O/r05_profile_round.log 2>&1
tail -3 $O/r05_profile_round.log
for c in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c18_prof$c -o p -- python benchmarks/config${c}_step.py --plain --steps 20 > /dev/null 2> $O/c18_prof$c.err
  cp $(find $O/c18_prof$c -name '*kernel_stats.csv' | head -1) $O/r05_config${c}_kernel_stats.csv
  python benchmarks/step_timeline.py $(find $O/c18_prof$c -name '*kernel_trace.csv' | head -1) > $O/r05_config${c}_timeline.txt
  rm -rf $O/c18_prof$c
done
timeout 300 python benchmarks/config5_step.py --dtype fp16
timeout 300 python benchmarks/config5_step.py --dtype fp32
timeout 300 python benchmarks/config4_step.py
timeout 300 python benchmarks/conv_split_ab.py --out $O/r05_conv_split_ab.json | tail -1
