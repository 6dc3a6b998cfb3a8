#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python benchmarks/layer_bisect.py --layer 0 --out $O/c2_bisect_l0.json > $O/c2_bisect_l0.log 2>&1
tail -3 $O/c2_bisect_l0.log
timeout 600 python benchmarks/layer_bisect.py --layer 3 --out $O/c2_bisect_l3.json > $O/c2_bisect_l3.log 2>&1
tail -3 $O/c2_bisect_l3.log
timeout 900 python -m pytest tests/test_msda_bordered_gpu.py -x -q > $O/c2_tests_a.log 2>&1
echo "tests_a rc=$?" | tee -a $O/c2_tests_a.log
tail -3 $O/c2_tests_a.log
