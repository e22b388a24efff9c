#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/phase_profile.py ${1:-standing} ${2:-10} 2048 2>&1 | tail -30
