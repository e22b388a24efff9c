#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 300 python scripts/phase_profile.py standing 10 2048 2>&1 | tail -17
timeout 300 python scripts/phase_profile.py walking 10 2048 2>&1 | tail -17
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['solver'], d.get('parity'))"
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --gait walking 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH walking', d['value'], d['solver'], d.get('parity'))"
