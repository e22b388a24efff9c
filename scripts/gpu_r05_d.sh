#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dispatch_order.py tests/test_gpu_solve.py tests/test_examples.py tests/test_gpu_tick_pipeline.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python scripts/dev/lpt_times.py 2>&1 | grep -v amdgpu.ids
cp hector_simulation_amd/libhector_mpc_hip.so /tmp/keep.so; cp hector_simulation_amd/libhector_mpc_hip.so.srchash /tmp/keep.hash
for F in "" "-DHMPC_BLOCK_MIN_NEW=2" "-DHMPC_BLOCK_MIN_NEW=4" "-DHMPC_BLOCK_MIN_NEW_3C=2" "-DHMPC_MFS_GT=3"; do
  echo "== flags: $F"
  HMPC_EXTRA_FLAGS="$F" timeout 900 python scripts/quick_times.py standing_b8192 standing_b1024 h20_single_b4096 3contact_b8192 3contact_b2048 2>&1 | grep -v amdgpu.ids
done
cp /tmp/keep.so hector_simulation_amd/libhector_mpc_hip.so; cp /tmp/keep.hash hector_simulation_amd/libhector_mpc_hip.so.srchash
