#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python scripts/xch_selftest.py 2>&1 | tail -8
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --check 4 2>&1 | tail -2 | cut -c1-600
