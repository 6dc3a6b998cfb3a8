"""Summarise a rocprofv3 kernel trace: per (kernel, grid) min / median / max duration in us.

    rocprofv3 --kernel-trace --output-format csv -d out -o p -- python benchmarks/head_micro.py
    python benchmarks/kernel_times.py out/p_kernel_trace.csv [name-substring]
"""
import csv
import sys
from collections import defaultdict


def main():
    needle = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        n = r["Kernel_Name"]
        if needle in n:
            key = (n.replace("void ", "").split("(")[0][-48:], r["Grid_Size_X"], r["Grid_Size_Y"])
            acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in acc.items():
        v.sort()
        print("%-50s grid %8s x %-3s n=%-4d min %8.1f  med %8.1f  max %8.1f" % (k[0], k[1], k[2], len(v), v[0], v[len(v) // 2], v[-1]))


if __name__ == "__main__":
    main()
