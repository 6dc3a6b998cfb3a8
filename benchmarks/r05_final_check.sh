#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --train-steps 0 --in-flight-report 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('BENCH', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v.get('images_per_s') for k,v in d.get('configs',{}).items()})"
