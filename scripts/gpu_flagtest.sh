#!/bin/bash
# developer: tests + bench of a build with extra compile flags ($1), product library restored afterwards
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cp hector_simulation_amd/libhector_mpc_hip.so /tmp/keep.so; cp hector_simulation_amd/libhector_mpc_hip.so.srchash /tmp/keep.hash
export HMPC_EXTRA_FLAGS="$1"
timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_contacts3.py tests/test_gpu_robustness.py tests/test_gpu_properties.py -m gpu -q 2>&1 | tail -12
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 64 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['solver'], d.get('parity'))
for k,v in d['other_configs'].items(): print(' ', k, v)"
timeout 300 python scripts/stress.py 2>&1 | tail -12
cp /tmp/keep.so hector_simulation_amd/libhector_mpc_hip.so; cp /tmp/keep.hash hector_simulation_amd/libhector_mpc_hip.so.srchash
