#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_neck_gpu.py tests/test_transformer_gpu.py tests/test_decoder_gpu.py tests/test_fp16_flavour_gpu.py -q -x > $O/c10_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c10_tests.log
tail -5 $O/c10_tests.log
timeout 300 python benchmarks/conv_split_ab.py --out $O/conv_split_ab.json
SDETR_CONV_SPLIT=1 timeout 300 python benchmarks/config5_step.py --dtype fp16
timeout 300 python benchmarks/config5_step.py --dtype fp16
timeout 300 python benchmarks/config5_step.py --dtype fp32
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c10_prof -o p -- python benchmarks/config5_step.py --plain --steps 20 > /dev/null 2> $O/c10_prof.err
f=$(find $O/c10_prof -name '*kernel_stats.csv' | head -1)
cp $f $O/c10_config5_kernel_stats.csv
python benchmarks/step_timeline.py $(find $O/c10_prof -name '*kernel_trace.csv' | head -1) > $O/c10_config5_timeline.txt
rm -rf $O/c10_prof
head -25 $O/c10_config5_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
