#!/bin/bash
# Kernel timeline of one graphed step of the headline bench on this box (rocprofv3 kernel trace):
#   [PRE='python statement'] bash benchmarks/timeline.sh NAME [grep pattern]   ->  gpurun_out/tl_NAME.txt (+ the matching lines)
# PRE runs after `import bench`, e.g. PRE='from salience_detr_amd import filter_ops; filter_ops.FFN_JOIN = False'.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -rf $O/tl_$1
rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1 -o p -- python -c "import sys; sys.argv = ['bench.py', '--plain', '--steps', '30', '--no-cpu-baseline']; import bench; ${PRE:-pass}; bench.main()" > /dev/null 2> $O/tl_$1.err
python benchmarks/step_timeline.py $(find $O/tl_$1 -name '*kernel_trace.csv' | head -1) > $O/tl_$1.txt
rm -rf $O/tl_$1
grep -E "${2:-.}" $O/tl_$1.txt
