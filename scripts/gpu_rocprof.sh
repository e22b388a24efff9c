#!/bin/bash
# rocprofv3 passes for profiles/: (1) kernel trace + stats, (2) PMC FETCH_SIZE, (3) PMC WRITE_SIZE, (4) SQ counters
# (separate passes: TCC slots do not fit both; counters never combined with other trace domains).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rocprof; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > gpurun_out/rocprof/bench_standing.json 2> gpurun_out/rocprof/bench.err
python bench.py --steps 20 --warmup 3 --gait walking --no-cpu-baseline > gpurun_out/rocprof/bench_walking.json 2>> gpurun_out/rocprof/bench.err
python bench.py --steps 10 --warmup 2 --horizon 20 --gait single --batch 4096 --no-cpu-baseline > gpurun_out/rocprof/bench_h20_single.json 2>> gpurun_out/rocprof/bench.err
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-configs --check 0 --streams 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rocprof/kt -o kt -- $CMD > gpurun_out/rocprof/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/rocprof/pmc_fetch -o pmc -- $CMD > gpurun_out/rocprof/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/rocprof/pmc_write -o pmc -- $CMD > gpurun_out/rocprof/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/rocprof/pmc_sq -o pmc -- $CMD > gpurun_out/rocprof/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/rocprof/pmc_sq2 -o pmc -- $CMD > gpurun_out/rocprof/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d gpurun_out/rocprof/pmc_f64 -o pmc -- $CMD > gpurun_out/rocprof/pmc_f64.log 2>&1
python scripts/phase_profile.py standing 10 6144 > gpurun_out/rocprof/phase_cycles.txt 2>/dev/null
python scripts/phase_profile.py walking 10 6144 >> gpurun_out/rocprof/phase_cycles.txt 2>/dev/null
python scripts/soak.py > gpurun_out/rocprof/soak.txt 2>&1
find gpurun_out/rocprof -name '*.db' -delete
cat gpurun_out/rocprof/bench_standing.json
head -3 gpurun_out/rocprof/kt/kt_kernel_stats.csv
