#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python benchmarks/image_lanes_probe.py --out $O/c6_lanes.json > $O/c6_lanes.log 2>&1
tail -4 $O/c6_lanes.log
timeout 900 python -m pytest tests/test_filter_ops_gpu.py tests/test_hotpath_gpu.py tests/test_training_step_full_gpu.py -q > $O/c6_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c6_tests.log
tail -4 $O/c6_tests.log
