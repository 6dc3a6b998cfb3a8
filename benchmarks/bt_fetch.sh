#!/bin/bash
# HBM fetch / write bytes of bt_main_kernel for scratch builds (benchmarks/bt_variant.sh): bash benchmarks/bt_fetch.sh NAME...
export TMPDIR=/tmp
O=$PWD/gpurun_out
for n in "$@"; do
  L=""; [ "$n" != product ] && L=benchmarks/libbt_$n.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/_btf
    LIB=$L rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/_btf -o p -- python benchmarks/msda_backward_ab.py --only-lds --queries 11363 --reps 5 > /dev/null 2>&1
    python - $n $c $(find $O/_btf -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys
n, c, f = sys.argv[1:4]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "bt_main" in r["Kernel_Name"] and r["Counter_Name"] == c]
print(n, c, "KiB per launch: %.0f (%d launches)" % (sum(v) / max(len(v), 1), len(v)))
PY
  done
done
rm -rf $O/_btf
