#!/bin/bash
# developer A/B: parity subset + kernel times + phase profile of the CURRENT build, then kernel times of builds with other -D flags
# usage: gpu_ab.sh "<quick_times cases>" "<phase profile args or ->" "<flags B>" "<flags C>" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
CASES="$1"; PH="$2"; shift; shift
timeout 900 python -m pytest tests/test_gpu_assembly.py tests/test_gpu_solve.py tests/test_gpu_contacts3.py tests/test_gpu_robustness.py -m gpu -x -q 2>&1 | tail -4
echo "== current build"
timeout 600 python scripts/quick_times.py $CASES 2>&1 | grep -v amdgpu.ids
if [ "$PH" != "-" ]; then timeout 300 python scripts/phase_profile.py $PH 2>/dev/null | head -24; fi
cp hector_simulation_amd/libhector_mpc_hip.so /tmp/keep.so; cp hector_simulation_amd/libhector_mpc_hip.so.srchash /tmp/keep.hash
for F in "$@"; do
  echo "== flags: $F"
  HMPC_EXTRA_FLAGS="$F" timeout 900 python scripts/quick_times.py $CASES 2>&1 | grep -v amdgpu.ids
done
cp /tmp/keep.so hector_simulation_amd/libhector_mpc_hip.so; cp /tmp/keep.hash hector_simulation_amd/libhector_mpc_hip.so.srchash
