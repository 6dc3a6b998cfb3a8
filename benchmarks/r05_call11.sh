#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_neck_gpu.py tests/test_transformer_gpu.py tests/test_fp16_flavour_gpu.py -q -x > $O/c11_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c11_tests.log
tail -5 $O/c11_tests.log
timeout 300 python benchmarks/conv_split_ab.py --out $O/conv_split_ab.json
timeout 300 python benchmarks/config5_step.py --dtype fp16
