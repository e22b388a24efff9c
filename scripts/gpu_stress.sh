#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cp /dev/null gpurun_out/stress.log
timeout 800 python scripts/stress.py 2>&1 | tail -30
