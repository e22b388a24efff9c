#!/bin/bash
# developer: the stand-alone stage-S prototype (scripts/micro/sweep_mfma64.hip) under several flag sets
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for F in "" "-DDESYNC" "-DLOOKAHEAD" "-DLOOKAHEAD -DDESYNC" "-DLOOKAHEAD -DDESYNC -DNO_MFMA" "$@"; do
  echo "== flags: $F"
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off $F scripts/micro/sweep_mfma64.hip -o /tmp/sweep_mfma64 2>/dev/null && timeout 120 /tmp/sweep_mfma64 2>&1 | grep -v amdgpu.ids
done
