#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_solve.py -x -q -m gpu 2>&1 | tail -2
for args in "" "--gait walking" "--horizon 20 --gait single --batch 4096"; do
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 8 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['config']['workload'][:40], round(d['value']), d['solver']['failed'], d['solver']['iters_median'], d['solver']['active_max'], d.get('parity'))"
done
