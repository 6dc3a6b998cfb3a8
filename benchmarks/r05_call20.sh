#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
timeout 300 python benchmarks/config5_step.py --dtype fp16 --library-linears
timeout 300 python benchmarks/config5_step.py --dtype fp16 --no-decoder-head
timeout 300 python benchmarks/config5_step.py --dtype fp16
done
