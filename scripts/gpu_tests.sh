#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
if [ -n "$1" ]; then
  echo "== again with $1"
  cp hector_simulation_amd/libhector_mpc_hip.so /tmp/keep.so; cp hector_simulation_amd/libhector_mpc_hip.so.srchash /tmp/keep.hash
  HMPC_EXTRA_FLAGS="$1" timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_contacts3.py tests/test_gpu_assembly.py -m gpu -x -q 2>&1 | tail -4
  cp /tmp/keep.so hector_simulation_amd/libhector_mpc_hip.so; cp /tmp/keep.hash hector_simulation_amd/libhector_mpc_hip.so.srchash
fi
