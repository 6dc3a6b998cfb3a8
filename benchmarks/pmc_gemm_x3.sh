cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/gx3
mkdir -p $O
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python benchmarks/gemm_x3_one.py ${GX3_ARGS:-22726 256 2048 y 5} > /dev/null 2> $O/p$i.err
done
python - <<'PY'
import csv, glob, collections
val=collections.defaultdict(list); dur=[]
for f in glob.glob('gpurun_out/gx3/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_x3' in r['Kernel_Name']: val[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('gpurun_out/gx3/p1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_x3' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('dur us', sum(dur)/len(dur), len(dur))
for k,v in sorted(val.items()): print(k, sum(v)/len(v))
PY
