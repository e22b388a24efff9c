#!/bin/bash
# rocprofv3 passes for profiles/: (1) kernel trace + stats, (2) PMC FETCH_SIZE, (3) PMC WRITE_SIZE (separate passes:
# TCC slots do not fit both; never combined with other trace domains).  Outputs under gpurun_out/rocprof/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rocprof; export TMPDIR=/tmp
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rocprof/kt -o kt -- $CMD > gpurun_out/rocprof/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/rocprof/pmc_fetch -o pmc -- $CMD > gpurun_out/rocprof/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/rocprof/pmc_write -o pmc -- $CMD > gpurun_out/rocprof/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/rocprof/pmc_sq -o pmc -- $CMD > gpurun_out/rocprof/pmc_sq.log 2>&1
find gpurun_out/rocprof -type f | head -40
for f in $(find gpurun_out/rocprof/kt -name '*kernel_stats*'); do head -5 $f; done
for f in $(find gpurun_out/rocprof/pmc_fetch -name '*counter_collection*'); do head -3 $f; done
# keep the merge small: drop the big traces, keep stats + counters
find gpurun_out/rocprof -name '*.db' -delete
