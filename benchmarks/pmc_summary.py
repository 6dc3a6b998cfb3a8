"""Mean per-launch value of every counter rocprofv3 collected for the sdetr:: MSDA kernels -> a markdown table.
usage: pmc_summary.py <dir prefix> <out.md> <nq> <batch> [kernel-name regex, default "msda"] [title]"""
import collections
import csv
import glob
import sys

import re
prefix, out, nq, batch = sys.argv[1:5]
pat = re.compile(sys.argv[5] if len(sys.argv) > 5 else "msda")
title = sys.argv[6] if len(sys.argv) > 6 else "fused MSDA forward kernels"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(prefix + "[0-9]*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat.search(k):
                agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat.search(k):
                dur[k.split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(out, "w") as fh:
    fh.write(f"# rocprofv3 --pmc passes, {title}, B={batch}, Nq={nq} (mean per launch)\n\n")
    for k in sorted(agg):
        ds = sorted(dur[k])
        fh.write(f"## {k}\n\nlaunch duration under the counter passes: median {ds[len(ds)//2]:.1f} us "
                 f"(min {ds[0]:.1f}, {len(ds)} launches)\n\n| counter | per launch |\n|---|---:|\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            fh.write(f"| {c} | {sum(v)/len(v):,.0f} |\n")
        fh.write("\n")
print(open(out).read())
