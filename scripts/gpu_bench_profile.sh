#!/bin/bash
# One gpurun call: smoke, GPU tests, bench lines, rocprofv3 kernel trace.  Outputs under gpurun_out/.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_standing.json 2> gpurun_out/bench_standing.err; cat gpurun_out/bench_standing.json; tail -3 gpurun_out/bench_standing.err
python bench.py --steps 20 --warmup 3 --gait walking --no-cpu-baseline > gpurun_out/bench_walking.json 2>> gpurun_out/bench_standing.err; cat gpurun_out/bench_walking.json
python bench.py --steps 5 --warmup 1 --batch 1024 --no-cpu-baseline > gpurun_out/bench_1024.json 2>> gpurun_out/bench_standing.err; cat gpurun_out/bench_1024.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 0 > gpurun_out/prof_kt.log 2>&1
ls -R gpurun_out/prof_kt | head -20
find gpurun_out/prof_kt -name '*kernel_stats*' | head -1 | xargs -r head -5
