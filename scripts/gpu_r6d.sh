#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_command_sweep.py tests/test_gpu_assembly.py tests/test_gpu_robustness.py -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 16 2>gpurun_out/bench_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['windows']['values'], d['solver']['failed'])
for k,v in d['other_configs'].items():
    if 'sweep' in k or 'range' in k or 'error' in k: print(' ', k, json.dumps(v))"
tail -3 gpurun_out/bench_err.txt
