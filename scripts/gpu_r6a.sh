#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_robustness.py -m gpu -x -q 2>&1 | tail -8
python scripts/dev/range_scale.py 2>&1 | grep -v amdgpu | tee gpurun_out/range_scale.txt
python scripts/quick_times.py standing_b8192 standing_b1024 h20_single_b4096 2>&1 | grep -v amdgpu | tee gpurun_out/quick.txt
