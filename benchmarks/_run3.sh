mkdir -p gpurun_out/r3e; O=gpurun_out/r3e
python -m pytest tests/test_filter_ops_gpu.py tests/test_hotpath_gpu.py tests/test_encoder_timed_mode_gpu.py tests/test_topk_attention_proj_gpu.py -x -q 2>&1 | tail -15 > $O/pytest.log
tail -4 $O/pytest.log
python bench.py --steps 30 --warmup 5 --in-flight-report 0 > $O/bench_hs.json 2> $O/bench_hs.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py --plain --steps 20 > $O/bench_profiled.json 2> $O/prof.err
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python benchmarks/step_timeline.py $(find $O/prof -name '*kernel_trace.csv' | head -1) > $O/step_timeline.txt
rm -rf $O/prof
for f in hs; do python -c "
import json,sys
d=json.load(open('$O/bench_$f.json')); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], r['frac'], r['frac_warm'], r['per_layer_us'])
print(d.get('parity_vs_cpu'))
"; done
grep -i "hsort\|prefilter\|topk_rank" $O/step_timeline.txt | head -30
tail -2 $O/step_timeline.txt
