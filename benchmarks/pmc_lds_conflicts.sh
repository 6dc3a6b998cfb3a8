cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ldsc; mkdir -p $O
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/train -o p -- python bench.py --mode train --no-graph --steps 2 --warmup 1 > /dev/null 2> $O/train.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/infer -o p -- python bench.py --plain --no-graph --steps 3 --warmup 2 > /dev/null 2> $O/infer.err
python - <<'PY'
import csv, glob, collections
for tag in ("train", "infer"):
    val = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"gpurun_out/ldsc/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_BUSY_CYCLES": cnt[k] += 1
    rows = []
    for k, c in val.items():
        busy = c.get("SQ_BUSY_CYCLES", 0) / 32
        conf, act = c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_IDX_ACTIVE", 0)
        if act > 0: rows.append((conf / 256, k, cnt[k], busy, conf / max(act, 1), conf / 256 / max(busy, 1)))
    rows.sort(reverse=True)
    print(tag, "(kernel, launches, busy cycles total, conflict share of LDS-active, conflict cycles per CU / busy cycles)")
    for conf, k, n, busy, share, frac in rows[:14]:
        print(f"  {k:70s} {n:5d} {busy:12.0f} {share:6.2f} {frac:6.3f}")
PY
