#!/bin/bash
# rocprofv3 counter passes over the two fused MSDA forward kernels at one layer size (GPU box, via gpurun).
#   bash benchmarks/pmc_msda.sh <tag> [NQ] [BATCH]   -> gpurun_out/<tag>_pmc_*/ , summarised by benchmarks/pmc_summary.py
# Counter passes carry --kernel-trace only (no other trace domain), one --pmc set per pass.
set -u
TAG=${1:-r02}
NQ=${2:-11363}
B=${3:-2}
O=$PWD/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
# (round 4: the round-3 resident kernel on plain maps and the bordered kernel without / with a row order, same operands)
CMD="python benchmarks/msda_bordered_ab.py --batch $B --nq $NQ --reps 5 --tiles 16 --minimal --out $O/${TAG}_pmc_ab.json"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${TAG}_pmc_$i -o p -- $CMD > /dev/null 2> $O/${TAG}_pmc_$i.err
done
python benchmarks/pmc_summary.py $O/${TAG}_pmc_ $O/${TAG}_pmc_summary.md "$NQ" "$B"
