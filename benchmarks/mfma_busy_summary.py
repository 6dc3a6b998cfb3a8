"""MFMA-busy summary of the dense kernels of the hot path from rocprofv3 --pmc passes over `bench.py --plain --no-graph`.

    python benchmarks/mfma_busy_summary.py <dir prefix of the passes> <out.md>

Per kernel (sdetr:: kernels that issue MFMAs): launches, mean launch duration (kernel trace of the same passes), the
utilisation of the matrix pipes -- SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES as rocprofv3 reports them, divided by 32:
the busy counter is summed over the 32 shader engines (checked on a kernel of known duration: SQ_BUSY_CYCLES =
32 x duration x clock), the MFMA counter over the 1024 SIMDs (one matrix pipe each), so the raw ratio is 1024 / 32 = 32
times the per-pipe utilisation -- and the flops the MFMAs
performed (SQ_INSTS_VALU_MFMA_MOPS_* x 512: the counters tally matrix operations in units of 512 flops at full EXEC) over
the launch duration against the dense peak (2.5 PFLOP/s bf16, 157 TFLOP/s f32-input MFMA; MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import sys

prefix, out = sys.argv[1:3]
val = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(prefix + "[0-9]*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sdetr::" in k:
                val[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sdetr::" in k:
                dur[k.split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def mean(v):
    return sum(v) / len(v) if v else 0.0


rows = []
for k in val:
    c = val[k]
    mfma_busy, busy = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), mean(c.get("SQ_BUSY_CYCLES", []))
    if mfma_busy <= 0:
        continue
    us = mean(dur[k])
    bf16 = mean(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [])) * 512
    f32 = mean(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", [])) * 512
    rows.append((us * len(dur[k]), k, len(dur[k]), us, mfma_busy, busy, bf16, f32))
rows.sort(reverse=True)
with open(out, "w") as fh:
    fh.write("# MFMA-busy counters of the dense kernels (rocprofv3 --pmc, bench.py --plain --no-graph, batch 2, bf16)\n\n")
    fh.write("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES) (means per launch; SQ_BUSY_CYCLES is summed over "
             "32 shader engines, the MFMA counter over 1024 SIMDs / matrix pipes).  Achieved = MFMA flops per launch "
             "(SQ_INSTS_VALU_MFMA_MOPS_{BF16,F32} x 512) / mean launch duration of the same passes; peak = 2500 TFLOP/s "
             "(bf16 MFMA) or 157 TFLOP/s (f32-input MFMA).  Durations under the counter passes run ~10-20 % above the "
             "un-profiled ones.\n\n")
    fh.write("| kernel | launches | mean us | MFMA busy | bf16 TFLOP/s (of 2500) | f32 TFLOP/s (of 157) |\n|---|---:|---:|---:|---:|---:|\n")
    for _, k, n, us, mb, b, bf16, f32 in rows:
        tb = bf16 / us / 1e6 if us else 0.0
        tf = f32 / us / 1e6 if us else 0.0
        fh.write(f"| `{k[:90]}` | {n} | {us:.1f} | {mb / b / 32 if b else 0:.3f} | "
                 f"{tb:.0f} ({tb / 2500:.2f}) | {tf:.1f} ({tf / 157:.2f}) |\n")
print(open(out).read())
