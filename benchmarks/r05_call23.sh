#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_mlp_rows_gpu.py tests/test_decoder_gpu.py tests/test_transformer_gpu.py tests/test_fp16_flavour_gpu.py -q -x > $O/c23_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c23_tests.log
tail -15 $O/c23_tests.log
for i in 1 2; do
timeout 300 python benchmarks/config5_step.py --dtype fp16 --no-decoder-head
timeout 300 python benchmarks/config5_step.py --dtype fp16
done
