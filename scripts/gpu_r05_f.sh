#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_contacts3.py tests/test_gpu_full_batch.py tests/test_gpu_robustness.py tests/test_gpu_tick_warm.py tests/test_examples.py -m gpu -x -q 2>&1 | tail -3
echo "== current build"
timeout 600 python scripts/quick_times.py 2>&1 | grep -v amdgpu.ids
cp hector_simulation_amd/libhector_mpc_hip.so /tmp/keep.so; cp hector_simulation_amd/libhector_mpc_hip.so.srchash /tmp/keep.hash
F="-DHMPC_S0_ACTIVE_ROWS=0"
echo "== flags: $F"
HMPC_EXTRA_FLAGS="$F" timeout 900 python scripts/quick_times.py 2>&1 | grep -v amdgpu.ids
cp /tmp/keep.so hector_simulation_amd/libhector_mpc_hip.so; cp /tmp/keep.hash hector_simulation_amd/libhector_mpc_hip.so.srchash
timeout 300 python scripts/phase_profile.py standing 10 6144 2>/dev/null | grep "blk:\|TOTAL\|sweep"
timeout 300 python scripts/phase_profile.py standing 10 2048 3 2>/dev/null | grep "blk:\|TOTAL\|sweep"
