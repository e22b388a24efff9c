#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/gpu_tests.txt
echo "== current build"
timeout 600 python scripts/quick_times.py 2>&1 | grep -v amdgpu.ids | tee $O/quick_times.txt
timeout 600 python scripts/dev/lpt_times.py 2>&1 | grep -v amdgpu.ids | tee $O/dispatch_order.txt
timeout 300 python scripts/phase_profile.py standing 10 2048 3 2>/dev/null | head -24 | tee $O/phase_3c.txt
cp hector_simulation_amd/libhector_mpc_hip.so /tmp/keep.so; cp hector_simulation_amd/libhector_mpc_hip.so.srchash /tmp/keep.hash
for F in "-DHMPC_SCHUR_MFMA_3C=0 -DHMPC_SCHUR_MFMA_WIDE=0"; do
  echo "== flags: $F"
  HMPC_EXTRA_FLAGS="$F" timeout 900 python scripts/quick_times.py 3contact_b2048 3contact_b8192 h20_double_b2048 h14_double_b2048 2>&1 | grep -v amdgpu.ids
done
cp /tmp/keep.so hector_simulation_amd/libhector_mpc_hip.so; cp /tmp/keep.hash hector_simulation_amd/libhector_mpc_hip.so.srchash
