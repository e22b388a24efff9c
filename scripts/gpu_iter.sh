#!/bin/bash
# developer loop: fast GPU parity subset, phase profile, short bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_contacts3.py tests/test_gpu_tick_warm.py tests/test_gpu_robustness.py -m gpu -x -q 2>&1 | tail -15
timeout 300 python scripts/phase_profile.py standing 10 2048 2>&1 | tail -21
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['solver'], d.get('parity'))
for k,v in d['other_configs'].items(): print(' ', k, v)"
