#!/bin/bash
# Scratch build of the library with ONE source compiled under extra -D flags (knock-outs, stamps, variants):
#   benchmarks/lib_variant.sh NAME SOURCE.hip [-DFLAG ...]  ->  benchmarks/libv_NAME.so (git-ignored)
# Every other object is the product build's.  Use with LIB=benchmarks/libv_NAME.so on the benchmark scripts that honour it.
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift; shift
python -c "from salience_detr_amd.csrc import build; build.build()"
C=salience_detr_amd/csrc
STEM=${SRC%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -munsafe-fp-atomics \
    "$@" -x hip -c $C/$SRC -o $C/_obj/$STEM.v_$NAME.o
OBJS=$(ls $C/_obj/*.o | grep -v '\.ablate\.o' | grep -v '\.v_' | grep -vE 'msda_backward_tiled\.[a-z0-9_]+\.o' | grep -v "/$STEM\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $C/_obj/$STEM.v_$NAME.o -o benchmarks/libv_$NAME.so
echo benchmarks/libv_$NAME.so
