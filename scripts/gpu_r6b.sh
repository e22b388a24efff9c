#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/dev/cont_probe.py 6 2>&1 | grep -v amdgpu
python scripts/dev/cont_probe.py 3 2>&1 | grep -v amdgpu
python scripts/dev/cont_probe.py 6 512 2>&1 | grep -v amdgpu
