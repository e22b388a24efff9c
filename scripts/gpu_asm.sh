#!/bin/bash
# developer: bit-exactness of the assembly stage + its phase profile (standing and walking)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_assembly.py tests/test_gpu_contacts3.py tests/test_gpu_solve.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python scripts/phase_profile.py standing 10 2048 2>&1 | grep -E "asm|TOTAL"
timeout 300 python scripts/phase_profile.py walking 10 2048 2>&1 | grep -E "asm|TOTAL"
