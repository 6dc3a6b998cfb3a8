#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_mlp_rows_gpu.py tests/test_decoder_gpu.py tests/test_transformer_gpu.py tests/test_fp16_flavour_gpu.py -q -x > $O/c21_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c21_tests.log
tail -15 $O/c21_tests.log
timeout 300 python benchmarks/config5_step.py --dtype fp16
timeout 300 python benchmarks/config5_step.py --dtype fp16
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c21_prof -o p -- python benchmarks/config5_step.py --plain --steps 20 > /dev/null 2> $O/c21_prof.err
f=$(find $O/c21_prof -name '*kernel_stats.csv' | head -1)
cp $f $O/c21_config5_kernel_stats.csv
rm -rf $O/c21_prof
grep "rows_linear\|mlp_rows\|decoder_head" $O/c21_config5_kernel_stats.csv | cut -c1-70,90-200
