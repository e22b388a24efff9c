#!/bin/bash
# A/B of the continuation variant's switches: what the continuation pass alone leaves (scripts/dev/cont_probe.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cp hector_simulation_amd/libhector_mpc_hip.so /tmp/keep.so; cp hector_simulation_amd/libhector_mpc_hip.so.srchash /tmp/keep.hash
for F in "" "$@"; do
  echo "== flags: '$F'"
  HMPC_ALLOW_DEV_BUILD=1 HMPC_EXTRA_FLAGS="$F" python scripts/dev/cont_probe.py 6 4096 2>&1 | grep -v "amdgpu\|hipcc"
done
cp /tmp/keep.so hector_simulation_amd/libhector_mpc_hip.so; cp /tmp/keep.hash hector_simulation_amd/libhector_mpc_hip.so.srchash
