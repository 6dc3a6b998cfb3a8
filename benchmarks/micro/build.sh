#!/bin/bash
# Builds the standalone micro-benchmarks for gfx950 next to their sources (cross-compiles without a GPU; the binaries
# are git-ignored and travel to the GPU box with the snapshot).      bash benchmarks/micro/build.sh
set -e
cd "$(dirname "$0")"
for f in lds_atomic_rate mfma_rate ffn_two_wave graph_launch_floor valu_rate gather_rate kernel_cold_start l2_prefetch hsort_phases mfma_valu_overlap; do
  [ -f $f.hip ] || continue
  extra=""
  [ $f = lds_atomic_rate ] && extra="-munsafe-fp-atomics"
  [ $f = hsort_phases ] && extra="-std=c++17 -DSDETR_HS_STAMPS -munsafe-fp-atomics"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $extra -o $f $f.hip 2> /dev/null && echo "built $f"
done
# csrc/gemm_x3.hip with one leg removed (see gemm_x3_ablate.hip)
for v in 0 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DSDETR_GX3_ABLATE=$v -I ../../include \
    -o gemm_x3_ablate_$v gemm_x3_ablate.hip 2> /dev/null && echo "built gemm_x3_ablate_$v"
done
# the library's benchmark build (ablated MSDA instantiations + phase stamps) for benchmarks/msda_bordered_ab.py --ablate / --stamps
python ../../salience_detr_amd/csrc/build.py --ablations > /dev/null && echo "built libsalience_hip_ablate.so"
