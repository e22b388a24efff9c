#!/bin/bash
# round 5, first GPU call: full GPU tests on the refactored build, the self-launching bench, baseline numbers, predictor data
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --check 64 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_gloo.err; echo "gpus2 rc=$?"; cut -c1-400 $O/bench_gpus2_gloo.json
timeout 120 python bench.py --gpus 8 --steps 1 > $O/bench_gpus8.out 2> $O/bench_gpus8.err; echo "gpus8 rc=$? (expected non-zero)"; tail -2 $O/bench_gpus8.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; cut -c1-300 $O/bench_default.json
timeout 600 python scripts/quick_times.py 2>&1 | grep -v amdgpu > $O/quick_times.txt; cat $O/quick_times.txt
timeout 600 python scripts/dev/predictor_data.py 2>&1 | grep -v amdgpu | tee $O/predictor_data.txt
timeout 300 python scripts/phase_profile.py standing 10 6144 2>/dev/null > $O/phase_standing.txt; head -30 $O/phase_standing.txt
