#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_neck_gpu.py tests/test_transformer_gpu.py tests/test_decoder_gpu.py tests/test_fp16_flavour_gpu.py -q -x > $O/c12_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c12_tests.log
tail -5 $O/c12_tests.log
timeout 300 python benchmarks/conv_split_ab.py --out $O/conv_split_ab.json | tail -1
timeout 300 python benchmarks/config5_step.py --dtype fp16
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c12_prof -o p -- python benchmarks/config5_step.py --plain --steps 20 > /dev/null 2> $O/c12_prof.err
f=$(find $O/c12_prof -name '*kernel_stats.csv' | head -1)
cp $f $O/c12_config5_kernel_stats.csv
python benchmarks/step_timeline.py $(find $O/c12_prof -name '*kernel_trace.csv' | head -1) > $O/c12_config5_timeline.txt
rm -rf $O/c12_prof
grep "se_gate\|grid_nms\|conv3x3" $O/c12_config5_kernel_stats.csv | cut -c1-60,120-260
