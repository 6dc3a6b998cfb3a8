"""Turn two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --plain --no-graph` into
profiles/rNN_msda_traffic.json: HBM bytes per fused-MSDA launch, per encoder layer.

    python benchmarks/pmc_to_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
        <bench.json> <out.json>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE is doubled per the gfx950 note in
MI355X_MICROARCH.md (16-byte-per-lane loads are tallied at half size); Infinity-Cache hits are included.
The fused-MSDA launches of an eager step come in layer order, so launch i of the pass belongs to layer i mod 6
(the coarse-levels-in-LDS kernel always launches one workgroup per CU: grid size no longer tells layers apart).
The file is tagged with the sha256 of the kernel sources it was measured on (bench.kernel_source_tag); bench.py
reports `traffic: null` when its sources differ.
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_layer_mean(path, counter, layers):
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if "msda_" in row["Kernel_Name"] and "Kernel" not in row["Kernel_Name"][:0] and row["Counter_Name"] == counter:
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0], float(row["Counter_Value"])))
    rows.sort()
    acc, names = defaultdict(list), {}
    for i, (_, name, v) in enumerate(rows):
        acc[i % layers].append(v)
        names[i % layers] = name
    if len(rows) % layers:
        raise SystemExit("%d MSDA launches are not a multiple of %d layers" % (len(rows), layers))
    return {l: sum(v) / len(v) for l, v in acc.items()}, names


def main():
    import bench
    fetch_csv, write_csv, bench_json, out = sys.argv[1:5]
    b = json.load(open(bench_json))
    nqs = b["roofline"]["num_queries_per_layer"]
    fetch, names = per_layer_mean(fetch_csv, "FETCH_SIZE", len(nqs))
    write, _ = per_layer_mean(write_csv, "WRITE_SIZE", len(nqs))
    per, by_layer = {}, []
    for l, n in enumerate(nqs):
        entry = {"FETCH_SIZE_KiB": round(fetch[l], 1), "WRITE_SIZE_KiB": round(write[l], 1),
                 "hbm_bytes": int((2 * fetch[l] + write[l]) * 1024), "kernel": names[l]}
        by_layer.append(dict(layer=l, num_query=n, **entry))
        per[str(n)] = entry           # layers with equal query counts: the later one (same kernel, same operand sizes)
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py "
                  "--plain --no-graph --steps 5 --warmup 3; benchmarks/pmc_to_traffic.py",
        "units": "counter values are KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per the "
                 "gfx950 note in MI355X_MICROARCH.md (16-byte-per-lane loads are tallied at half size); "
                 "Infinity-Cache hits are included in FETCH_SIZE",
        "kernel_source_tag": bench.kernel_source_tag(), "batch": b["config"]["batch_per_gpu"], "dtype": b["dtype"],
        "value_dtype": b["config"]["value_map_storage"], "per_num_query": per, "per_layer": by_layer,
    }, open(out, "w"), indent=1)
    print(json.dumps(by_layer))


if __name__ == "__main__":
    main()
