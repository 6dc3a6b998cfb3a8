mkdir -p gpurun_out/r3f; O=gpurun_out/r3f
./benchmarks/micro/hsort_phases > $O/hsort_phases.json 2>&1; cat $O/hsort_phases.json
python -m pytest tests/test_filter_ops_gpu.py tests/test_hotpath_gpu.py -x -q 2>&1 | tail -4
python bench.py --steps 30 --warmup 5 --in-flight-report 0 --no-cpu-baseline > $O/bench_hs.json 2> $O/bench_hs.err
python -c "
import json,sys
d=json.load(open('$O/bench_hs.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['frac_warm'], r['per_layer_us'])
"
