#!/bin/bash
# Runs on the GPU box (gpurun): the official bench line, its rocprofv3 kernel summary + step timeline, the PMC passes
# behind roofline.traffic, the MSDA kernel counters, the MFMA-busy counters of the dense kernels, the kernel A/B, the
# training step and the standalone micro-benchmarks.      bash benchmarks/profile_round.sh r03
# Everything lands in gpurun_out/<tag>_*; copy what should be judged into profiles/.
# Counter passes carry --kernel-trace only (no other trace domain), one --pmc set per pass.
# (The survey's full CPU-baseline protocol -- minutes of host time -- runs last when FULL_CPU=1:
# `python bench.py --cpu-protocol full` -> <tag>_bench_cpu_full.json.)
set -u
R=${1:-r06}
O=$PWD/gpurun_out
mkdir -p $O profiles
export TMPDIR=/tmp
# 1. PMC traffic passes first: the bench line's roofline.traffic is read from profiles/${R}_msda_traffic.json
python bench.py --no-cpu-baseline --train-steps 0 --in-flight-report 0 --config-steps 0 > $O/${R}_bench_quick.json 2> $O/${R}_bench_quick.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${R}_pmc_$c -o p -- \
    python bench.py --plain --no-graph --steps 5 --warmup 3 > /dev/null 2> $O/${R}_pmc_$c.err
done
F=$(find $O/${R}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $O/${R}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python benchmarks/pmc_to_traffic.py $F $W $O/${R}_bench_quick.json $O/${R}_msda_traffic.json && \
  cp $O/${R}_msda_traffic.json profiles/${R}_msda_traffic.json      # (this box's copy: what the bench run below reads)
rm -rf $O/${R}_pmc_FETCH_SIZE $O/${R}_pmc_WRITE_SIZE
#    ... and the backward op's counters (the train_step record reads profiles/${R}_msda_bwd_traffic.json)
bash benchmarks/pmc_msda_bwd.sh ${R} 11363 2 > /dev/null 2>&1
cp $O/${R}_msda_bwd_traffic.json profiles/${R}_msda_bwd_traffic.json 2> /dev/null
# 2. the kernel summary of the plain timed loop (six layers in equal proportion) FIRST: the bench line's
#    roofline.timing_rocprof_us is read from profiles/${R}_msda_rocprof.json; then the official bench line (with the
#    train_step sub-record, the config sub-records and the CPU baseline)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof -o p -- python bench.py --plain --steps 50 > $O/${R}_bench_profiled.json 2> $O/${R}_prof.err
cp $(find $O/${R}_prof -name '*kernel_stats.csv' | head -1) $O/${R}_bench_kernel_stats.csv
python benchmarks/step_timeline.py $(find $O/${R}_prof -name '*kernel_trace.csv' | head -1) > $O/${R}_step_timeline.txt
python - $(find $O/${R}_prof -name '*kernel_trace.csv' | head -1) $O/${R}_msda_rocprof.json <<'PY'
import csv, json, sys
sys.path.insert(0, ".")
import bench
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "msda_bordered_kernel" in r["Kernel_Name"] or "msda_resident_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
per_layer = [round(sum(us[k::6]) / len(us[k::6]), 2) for k in range(6)] if len(us) % 6 == 0 else None
json.dump({"source": "rocprofv3 --kernel-trace over `python bench.py --plain --steps 50` (hipGraph replay): every fused-MSDA launch of "
                     "the timed loop and its warm-up, launch i of the trace = encoder layer i mod 6",
           "kernel": rows[0]["Kernel_Name"].split("(")[0], "launches": len(us), "avg_launch_us": round(sum(us) / len(us), 2),
           "per_layer_us": per_layer, "batch": 2, "kernel_source_tag": bench.kernel_source_tag()}, open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
PY
cp $O/${R}_msda_rocprof.json profiles/${R}_msda_rocprof.json
rm -rf $O/${R}_prof
python bench.py --steps 20 --warmup 5 > $O/${R}_bench.json 2> $O/${R}_bench.err
tail -c 300 $O/${R}_bench.json
head -8 $O/${R}_bench_kernel_stats.csv | cut -c1-150
# 3. counters of the fused MSDA forward kernels (layer 0 size: round 3's resident kernel, the bordered kernel without and
#    with a row order) and of the MSDA backward kernels
bash benchmarks/pmc_msda.sh ${R} 11363 2 > /dev/null 2>&1
mv $O/${R}_pmc_summary.md $O/${R}_msda_pmc.md 2> /dev/null
#    the shader clock the chip holds under the gather: SQ_BUSY_CYCLES (summed over the 32 shader engines' sequencers) / 32 /
#    launch duration of the same pass, for the bordered kernel in tile order (what the step launches) -> roofline.valu
python - $O/${R}_pmc_1 $O/${R}_msda_clock.json <<'PY'
import csv, glob, json, sys
sys.path.insert(0, ".")
import bench
d, out = sys.argv[1:3]
busy, dur = {}, {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_bordered_kernel<false, 2, false, true" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_BUSY_CYCLES":
            busy[r["Dispatch_Id"]] = busy.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_bordered_kernel<false, 2, false, true" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ghz = sorted(busy[k] / 32.0 / dur[k] for k in busy if k in dur and dur[k] > 0)
if ghz:
    json.dump({"shader_clock_ghz": round(ghz[len(ghz) // 2], 3), "launches": len(ghz), "kernel_source_tag": bench.kernel_source_tag(),
               "source": "median over the tile-order bordered-kernel launches of SQ_BUSY_CYCLES / 32 sequencers / launch duration, "
                         "rocprofv3 --pmc pass at 11 363 queries (benchmarks/pmc_msda.sh)"},
              open(out, "w"), indent=1)
    print(open(out).read())
PY
cp $O/${R}_msda_clock.json profiles/${R}_msda_clock.json 2> /dev/null
rm -rf $O/${R}_pmc_[1-5]
# 4. MFMA-busy counters of the dense kernels
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d $O/${R}_mfma_1 -o p -- \
  python bench.py --plain --no-graph --steps 5 --warmup 3 > /dev/null 2> $O/${R}_mfma_1.err
python benchmarks/mfma_busy_summary.py $O/${R}_mfma_ $O/${R}_mfma_busy.md > /dev/null
rm -rf $O/${R}_mfma_1
# 5. kernel A/B at batch 2 and at batch 16 (value maps beyond the 256 MiB Infinity Cache): round 3's resident kernel, the
#    bordered kernel in list / raster / tile order; then on the step's own operands with the ablations and phase stamps
python benchmarks/msda_bordered_ab.py --tiles 8,16,32 --out $O/${R}_msda_ab.json > /dev/null 2>&1
python benchmarks/msda_bordered_ab.py --batch 16 --reps 10 --tiles 16 --nq 11363,6817,4545 --out $O/${R}_msda_ab_b16.json > /dev/null 2>&1
python benchmarks/msda_bordered_ab.py --levels 5scale --batch 1 --nq 45330,27198,9066 --tiles 16 --out $O/${R}_msda_ab_5scale_bordered.json > /dev/null 2>&1
python benchmarks/msda_real_operands.py --ablate 1,2,3,12,15,16,64,2048,128,256,1024 --stamps --tiles 8,32 --out $O/${R}_msda_real_operands.json > /dev/null 2>&1
python benchmarks/msda_backward_ab.py > $O/${R}_msda_backward_ab.json 2> /dev/null
#    ... and on the reference's 5scale pyramid (level 3 alone resident), one and two images
python benchmarks/msda_resident_ab.py --levels 5scale --batch 1 --nq 45330,36264,27198,18132,9066 --chunks 0 --out $O/${R}_msda_ab_5scale_b1.json > /dev/null 2>&1
python benchmarks/msda_resident_ab.py --levels 5scale --batch 2 --nq 45330,9066 --chunks 0 --out $O/${R}_msda_ab_5scale_b2.json > /dev/null 2>&1
python - <<PY
import json
a = json.load(open("$O/${R}_msda_ab_5scale_b1.json")); b = json.load(open("$O/${R}_msda_ab_5scale_b2.json"))
json.dump({"note": "benchmarks/msda_resident_ab.py --levels 5scale (level 3 alone resident in LDS); batch 1 rows then batch 2 rows",
           "rows": a + b}, open("$O/${R}_msda_ab_5scale.json", "w"), indent=1)
PY
rm -f $O/${R}_msda_ab_5scale_b1.json $O/${R}_msda_ab_5scale_b2.json
# 6. training step, for the backward kernel
python bench.py --mode train --steps 5 --warmup 2 > $O/${R}_train.json 2> $O/${R}_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_train -o p -- python bench.py --mode train --steps 3 --warmup 1 > /dev/null 2> $O/${R}_prof_train.err
cp $(find $O/${R}_prof_train -name '*kernel_stats.csv' | head -1) $O/${R}_train_kernel_stats.csv
rm -rf $O/${R}_prof_train
#    the fp32-accurate GEMM on the step's Linear shapes, both generations against the library
python benchmarks/gemm_x3_bench.py > $O/${R}_gemm_x3.json 2> /dev/null
python benchmarks/train_op_profile.py > $O/${R}_train_ops.txt 2> /dev/null
python benchmarks/gemm_epilogue_probe.py --out $O/${R}_gemm_epilogue.json > /dev/null 2>&1
# 7. standalone micro-benchmarks of this round (built by benchmarks/micro/build.sh)
for m in l2_prefetch hsort_phases mfma_valu_overlap; do
  [ -x benchmarks/micro/$m ] && timeout 60 ./benchmarks/micro/$m > $O/${R}_$m.json 2> /dev/null
done
( echo '{"gemm_x3_ablate": ['
  for v in 0 1 2 3 4 5 6; do
    [ -x benchmarks/micro/gemm_x3_ablate_$v ] && timeout 60 ./benchmarks/micro/gemm_x3_ablate_$v | sed 's/$/,/'
  done
  for K in 64 1024 2048; do timeout 60 ./benchmarks/micro/gemm_x3_ablate_0 22726 $K 2048 | sed 's/$/,/'; done
  timeout 60 ./benchmarks/micro/gemm_x3_ablate_0 22726 256 2048 fwd 1 | sed 's/}$/, "generation": "128x128 tiles"}/'
  echo ']}' ) > $O/${R}_gemm_x3_ablate.json 2> /dev/null
if [ "${FULL_CPU:-0}" = "1" ]; then
  python bench.py --cpu-protocol full --train-steps 0 --in-flight-report 0 --config-steps 0 > $O/${R}_bench_cpu_full.json 2> $O/${R}_bench_cpu_full.err
fi
ls $O | grep "^${R}" | head -40
