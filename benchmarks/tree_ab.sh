#!/bin/bash
# Same-box A/B of two TREES on the headline step: the round-5 tree (git archive of the round-5 commit, built, under
# _r05tree/ -- git-ignored) against this tree, `bench.py --plain --steps 50` in turns.   bash benchmarks/tree_ab.sh [pairs]
P=${1:-3}
R=$PWD
for i in $(seq $P); do
  for t in _r05tree .; do
    cd $R/$t
    v=$(python bench.py --plain --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "$t $v"
  done
done
