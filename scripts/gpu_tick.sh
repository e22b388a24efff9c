#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tick_warm.py tests/test_gpu_contacts3.py -m gpu -x -q -s 2>&1 | tail -40 | tee gpurun_out/tick.log
timeout 600 python -m pytest tests/ -m gpu -x -q --deselect tests/test_gpu_contacts3.py --deselect tests/test_gpu_tick_warm.py 2>&1 | tail -5
