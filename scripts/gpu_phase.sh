#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/phase_profile.py standing 10 2048 2>&1 | tail -22
