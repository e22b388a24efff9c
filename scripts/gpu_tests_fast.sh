#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_properties.py::test_full_size_properties 2>&1 | tail -8
bash scripts/gpu_quick2.sh 2>&1 | grep BENCH
