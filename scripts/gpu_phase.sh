#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/phase_profile.py standing 10 2048 2>&1 | tail -20
python scripts/phase_profile.py walking 10 2048 2>&1 | tail -20
