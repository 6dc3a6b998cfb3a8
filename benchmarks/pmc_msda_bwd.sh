#!/bin/bash
# rocprofv3 counter passes over the MSDA BACKWARD kernels (LDS-accumulating bt_main_kernel + bucketing, and the direct
# global-atomic kernel) at one layer size (GPU box, via gpurun).
#   bash benchmarks/pmc_msda_bwd.sh <tag> [NQ] [BATCH]   -> gpurun_out/<tag>_msda_bwd_pmc.md, <tag>_msda_bwd_traffic.json
# Counter passes carry --kernel-trace only (no other trace domain), one --pmc set per pass.
set -u
TAG=${1:-r04}
NQ=${2:-11363}
B=${3:-2}
O=$PWD/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
CMD="python benchmarks/msda_backward_ab.py --batch $B --queries $NQ --reps 5"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VMEM" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${TAG}_bwdpmc_$i -o p -- $CMD > /dev/null 2> $O/${TAG}_bwdpmc_$i.err
done
SDETR_TRAFFIC_SOURCES="salience_detr_amd/csrc/msda_backward_tiled.hip salience_detr_amd/csrc/msda_backward.hip" SDETR_TRAFFIC_OP_REGEX="bt_" \
  python benchmarks/pmc_summary.py $O/${TAG}_bwdpmc_ $O/${TAG}_msda_bwd_pmc.md "$NQ" "$B" "bt_main|bt_tile|bt_order|bt_clear|msda_col2im" "MSDA backward kernels (reference layout, fp32)" $O/${TAG}_msda_bwd_traffic.json > /dev/null
rm -rf $O/${TAG}_bwdpmc_[1-6]
