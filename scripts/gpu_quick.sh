#!/bin/bash
# developer: bitwise assembly + solve parity, then kernel times of the BASELINE shapes (no CPU baseline, no side legs)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_assembly.py tests/test_gpu_solve.py tests/test_gpu_contacts3.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python scripts/quick_times.py "$@" 2>&1 | tail -12
