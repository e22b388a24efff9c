#!/bin/bash
# developer: kernel times of the BASELINE shapes for several builds differing in -D flags.
# usage: gpu_flags.sh "<cases>" "<flags A>" "<flags B>" ...   (the product library is restored afterwards)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cp hector_simulation_amd/libhector_mpc_hip.so /tmp/keep.so; cp hector_simulation_amd/libhector_mpc_hip.so.srchash /tmp/keep.hash
CASES="$1"; shift
for F in "$@"; do
  echo "== flags: $F"
  HMPC_EXTRA_FLAGS="$F" timeout 900 python scripts/quick_times.py $CASES 2>&1 | grep -v amdgpu.ids
done
cp /tmp/keep.so hector_simulation_amd/libhector_mpc_hip.so; cp /tmp/keep.hash hector_simulation_amd/libhector_mpc_hip.so.srchash
