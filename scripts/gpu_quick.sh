#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for args in "--batch 32768" "--batch 32768 --gait mixed" "--batch 16384 --horizon 20 --gait single" "--batch 32768 --gait walking"; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 8 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['config']['workload'], d['value'], d['solver'], d.get('parity'))"
done
