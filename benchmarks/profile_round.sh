#!/bin/bash
# Runs on the GPU box (gpurun): the official bench line, its rocprofv3 kernel summary, and the PMC traffic passes.
#   bash benchmarks/profile_round.sh r01      -> gpurun_out/r01_*   (copy what should be judged into profiles/)
set -u
R=${1:-r01}
O=$PWD/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
# 1. PMC traffic passes first: the bench line's roofline.traffic is read from profiles/${R%%x}_msda_traffic.json
python bench.py --no-cpu-baseline --steps 10 --warmup 5 > $O/${R}_bench_quick.json 2> $O/${R}_bench_quick.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${R}_pmc_$c -o p -- \
    python bench.py --no-cpu-baseline --no-graph --steps 5 --warmup 3 --instrumented-steps 2 > /dev/null 2> $O/${R}_pmc_$c.err
done
F=$(find $O/${R}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $O/${R}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python benchmarks/pmc_to_traffic.py $F $W $O/${R}_bench_quick.json $O/${R}_msda_traffic.json
cp $O/${R}_msda_traffic.json profiles/r01_msda_traffic.json      # (this box's copy: what the bench run below reads)
# 2. the official bench line and its kernel summary
python bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err
tail -c 600 $O/${R}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof -o p -- python bench.py --no-cpu-baseline > $O/${R}_bench_profiled.json 2> $O/${R}_prof.err
S=$(find $O/${R}_prof -name '*kernel_stats.csv' | head -1)
cp $S $O/${R}_bench_kernel_stats.csv
head -25 $O/${R}_bench_kernel_stats.csv | cut -c1-150
# training step, for the backward kernel
python bench.py --mode train --steps 5 --warmup 2 > $O/${R}_train.json 2> $O/${R}_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_train -o p -- python bench.py --mode train --steps 3 --warmup 1 > /dev/null 2> $O/${R}_prof_train.err
S=$(find $O/${R}_prof_train -name '*kernel_stats.csv' | head -1)
cp $S $O/${R}_train_kernel_stats.csv
head -25 $O/${R}_train_kernel_stats.csv | cut -c1-150
