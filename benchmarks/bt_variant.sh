#!/bin/bash
# Scratch build of the library with msda_backward_tiled.hip compiled under extra -D flags (knock-outs / variants of the LDS
# backward): benchmarks/bt_variant.sh NAME [-DFLAG ...] -> benchmarks/libbt_NAME.so (git-ignored).  Every other object is the
# product build's.  Use with: python benchmarks/msda_backward_ab.py --lib benchmarks/libbt_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
python -c "from salience_detr_amd.csrc import build; build.build()"
C=salience_detr_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -munsafe-fp-atomics \
    "$@" -x hip -c $C/msda_backward_tiled.hip -o $C/_obj/msda_backward_tiled.$NAME.o
OBJS=$(ls $C/_obj/*.o | grep -v '\.ablate\.o' | grep -v 'msda_backward_tiled\.' )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $C/_obj/msda_backward_tiled.$NAME.o -o benchmarks/libbt_$NAME.so
echo benchmarks/libbt_$NAME.so
