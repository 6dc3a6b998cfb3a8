#!/bin/bash
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/c16_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/c16_tests.log
tail -5 $O/c16_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
