#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_transformer_gpu.py tests/test_decoder_gpu.py tests/test_fp16_flavour_gpu.py tests/test_graph_lanes_gpu.py -q -x > $O/c14_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c14_tests.log
tail -5 $O/c14_tests.log
timeout 300 python benchmarks/config5_step.py --dtype fp16 --no-overlap
timeout 300 python benchmarks/config5_step.py --dtype fp16
timeout 300 python benchmarks/config5_step.py --dtype fp16 --no-overlap
timeout 300 python benchmarks/config5_step.py --dtype fp16
