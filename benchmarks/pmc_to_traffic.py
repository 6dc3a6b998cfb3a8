"""Turn two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/rNN_msda_traffic.json.

    python benchmarks/pmc_to_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
        <bench.json> <out.json> [kernel-name-substring]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE is doubled per the gfx950 note in
MI355X_MICROARCH.md (16-byte-per-lane loads are tallied at half size); Infinity-Cache hits are included.
Launches are matched to encoder layers by grid size (larger grid = more queries).
"""
import csv
import json
import sys
from collections import defaultdict


def per_grid_mean(path, counter, needle):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle in row["Kernel_Name"] and row["Counter_Name"] == counter:
                acc[int(row["Grid_Size"])].append(float(row["Counter_Value"]))
    return {g: sum(v) / len(v) for g, v in acc.items()}


def main():
    fetch_csv, write_csv, bench_json, out = sys.argv[1:5]
    needle = sys.argv[5] if len(sys.argv) > 5 else "msda_gather"
    bench = json.load(open(bench_json))
    nqs = sorted(set(bench["roofline"]["num_queries_per_layer"]), reverse=True)
    fetch = per_grid_mean(fetch_csv, "FETCH_SIZE", needle)
    write = per_grid_mean(write_csv, "WRITE_SIZE", needle)
    grids = sorted(fetch, reverse=True)
    if len(grids) != len(nqs) or sorted(write, reverse=True) != grids:
        raise SystemExit("grid sizes %s / %s do not match query counts %s" % (grids, sorted(write), nqs))
    per = {}
    for g, n in zip(grids, nqs):
        per[str(n)] = {"FETCH_SIZE_KiB": round(fetch[g], 1), "WRITE_SIZE_KiB": round(write[g], 1),
                       "hbm_bytes": int((2 * fetch[g] + write[g]) * 1024)}
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py "
                  "--no-cpu-baseline --no-graph --steps 5 --warmup 3 --instrumented-steps 2; benchmarks/pmc_to_traffic.py",
        "units": "counter values are KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per the "
                 "gfx950 note in MI355X_MICROARCH.md (16-byte-per-lane loads are tallied at half size); "
                 "Infinity-Cache hits are included in FETCH_SIZE",
        "kernel_match": needle, "batch": bench["config"]["batch_per_gpu"], "dtype": bench["dtype"],
        "value_dtype": bench["config"]["value_map_storage"], "per_num_query": per,
    }, open(out, "w"), indent=1)
    print(json.dumps(per))


if __name__ == "__main__":
    main()
