"""Mean per-launch value of every counter rocprofv3 collected for the sdetr:: MSDA kernels -> a markdown table.
usage: pmc_summary.py <dir prefix> <out.md> <nq> <batch> [kernel-name regex, default "msda"] [title] [traffic.json]
traffic.json (optional): HBM bytes per launch of every matched kernel, (2 * FETCH_SIZE + WRITE_SIZE) KiB as
pmc_to_traffic.py derives them, tagged with the sha256 of the source files named in SDETR_TRAFFIC_SOURCES."""
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

if len(sys.argv) > 7:
    import hashlib
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    srcs = os.environ.get("SDETR_TRAFFIC_SOURCES", "").split()
    for f in srcs:
        h.update(open(os.path.join(root, f), "rb").read())
    per = {}
    for k in sorted(agg):
        if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
            f_, w_ = (sum(agg[k][c]) / len(agg[k][c]) for c in ("FETCH_SIZE", "WRITE_SIZE"))
            ds = sorted(dur[k])
            per[k] = {"FETCH_SIZE_KiB": round(f_, 1), "WRITE_SIZE_KiB": round(w_, 1), "hbm_bytes": int((2 * f_ + w_) * 1024),
                      "median_us_under_counters": round(ds[len(ds) // 2], 1)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace; benchmarks/pmc_summary.py",
               "units": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 note in MI355X_MICROARCH.md",
               "num_query": int(nq), "batch": int(batch), "sources": srcs, "source_tag": h.hexdigest()[:16],
               "per_kernel": per, "op_kernels": os.environ.get("SDETR_TRAFFIC_OP_REGEX", ".*"),
               "hbm_bytes_per_op": sum(v["hbm_bytes"] for k, v in per.items()
                                       if re.search(os.environ.get("SDETR_TRAFFIC_OP_REGEX", ".*"), k))},
              open(sys.argv[7], "w"), indent=1)
