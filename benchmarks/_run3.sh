mkdir -p gpurun_out/r3j; O=gpurun_out/r3j
python -m pytest tests/test_msda_timed_kernels_gpu.py tests/test_msda_gpu.py tests/test_filter_ops_gpu.py -x -q 2>&1 | tail -4
python -m pytest tests/test_hotpath_gpu.py tests/test_encoder_timed_mode_gpu.py tests/test_decoder_gpu.py tests/test_transformer_gpu.py -x -q 2>&1 | tail -4
python bench.py --steps 30 --warmup 5 --in-flight-report 0 --train-steps 0 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json,sys
d=json.load(open('$O/bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['frac_warm'], r['per_layer_us'], r['per_layer_us_warm'])
"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py --plain --steps 20 > $O/bench_profiled.json 2> $O/prof.err
python benchmarks/step_timeline.py $(find $O/prof -name '*kernel_trace.csv' | head -1) > $O/step_timeline.txt
rm -rf $O/prof
grep -i "msda_res\|merge_sorted" $O/step_timeline.txt | cut -c1-110; tail -1 $O/step_timeline.txt
