#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['solver'], d.get('parity'))
for k,v in d['other_configs'].items(): print(' ', k, v)"
