#!/bin/bash
# Same-box sweep of the hoisted head's carriers (benchmarks/hoist_ab.py): "FIN_LEVEL PARTS" per line of the list below.
cd "$(dirname "$0")/.."
for cfg in "1 3,3" "hoist 3,3" "2 3,3" "hoist 2,2,2" "hoist 3,2,1"; do
  set -- $cfg
  echo "== FIN_LEVEL=$1 PARTS=$2"
  FIN_LEVEL=$1 PARTS=$2 timeout 300 python benchmarks/hoist_ab.py 2>&1 | tail -1
done
