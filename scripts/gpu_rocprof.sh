#!/bin/bash
# rocprofv3 passes for profiles/<round>/ (one gpurun call).  Passes are kept separate: kernel trace + stats, then one
# --pmc pass per counter group (TCC slots do not fit both FETCH and WRITE; counters are never combined with other trace
# domains).  scripts/summarize_rocprof.py turns the output into the tracked summaries.
#  (1) bench lines: headline (default two launch streams, with the CPU baseline), walking, h=20 single support (run last, see below)
#  (2) headline workload, --streams 1: kernel trace + stats, PMC passes (HBM traffic, SQ counters, fp64 instruction mix)
#  (3) headline workload, DEFAULT two streams: kernel trace (start/end of consecutive dispatches: the overlap the value uses)
#  (4) every other kernel variant (walking 60 variables, h=20 single support, three contacts, wide double support):
#      kernel trace + stats and one SQ counter pass each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rocprof; rm -rf gpurun_out/rocprof/*; export TMPDIR=/tmp
O=gpurun_out/rocprof
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-configs --check 0 --streams 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- $CMD > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o pmc -- $CMD > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o pmc -- $CMD > $O/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $O/pmc_f64 -o pmc -- $CMD > $O/pmc_f64.log 2>&1
# (3) the default two-stream mode, trace only
rocprofv3 --kernel-trace --output-format csv -d $O/kt2 -o kt2 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-configs --check 0 > $O/kt2.log 2>&1
# (4) the other variants
for CASE in walking_b8192 h20_single_b4096 3contact_b8192 3contact_b2048 h20_double_b2048; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/var_$CASE/kt -o kt -- python scripts/quick_times.py $CASE > $O/var_$CASE.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/var_$CASE/pmc -o pmc -- python scripts/quick_times.py $CASE >> $O/var_$CASE.log 2>&1
done
python scripts/phase_profile.py standing 10 6144 > $O/phase_cycles.txt 2>/dev/null
python scripts/phase_profile.py walking 10 6144 >> $O/phase_cycles.txt 2>/dev/null
python scripts/phase_profile.py single 20 4096 >> $O/phase_cycles.txt 2>/dev/null
python scripts/phase_profile.py standing 10 2048 3 >> $O/phase_cycles.txt 2>/dev/null
python scripts/dev/latency_vs_batch.py 2>/dev/null | grep -v amdgpu > $O/latency_vs_batch.txt
python scripts/dev/lpt_times.py 2>/dev/null | grep -v amdgpu > $O/dispatch_order.txt
python scripts/soak.py > $O/soak.txt 2>&1
python scripts/stress.py 2>/dev/null | grep -v amdgpu > $O/stress.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/gpu_tests.txt
# (1) last: the bench lines read profiles/hbm_traffic.json, which only counts for the build it was taken on -- refresh it
# from the PMC passes above first (on this box's copy of the tree; the caller runs the summary again on its own copy)
ROUND=${1:-r06}
python scripts/summarize_rocprof.py $ROUND > /dev/null 2>&1
python bench.py --steps 20 --warmup 3 > $O/bench_standing.json 2> $O/bench.err
python bench.py --steps 20 --warmup 3 --gait walking --no-cpu-baseline > $O/bench_walking.json 2>> $O/bench.err
python bench.py --steps 10 --warmup 2 --horizon 20 --gait single --batch 4096 --no-cpu-baseline > $O/bench_h20_single.json 2>> $O/bench.err
# the N > 1 path launched by bench.py itself (two ranks on the one GPU over the gloo TEST transport), and the refusal of --gpus 8
python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --check 64 > $O/bench_gpus2_gloo.json 2>> $O/bench.err
python bench.py --gpus 8 --steps 1 > /dev/null 2> $O/bench_gpus8_refused.txt; echo "exit code $?" >> $O/bench_gpus8_refused.txt
find $O -name '*.db' -delete
find $O -name '*_agent_info.csv' -delete
cat $O/bench_standing.json | cut -c1-600
head -3 $O/kt/kt_kernel_stats.csv
ls $O
# round 6: what an off-nominal batch costs (fast pass, device repair with / without the hand-over), what the continuation pass alone leaves
python scripts/dev/range_scale.py 2>/dev/null | grep -v amdgpu > $O/range_scale.txt
python scripts/dev/range_scale.py 4096 single 20 2>/dev/null | grep -v amdgpu >> $O/range_scale.txt
python scripts/dev/range_scale.py 8192 mixed 10 2>/dev/null | grep -v amdgpu >> $O/range_scale.txt
( python scripts/dev/cont_probe.py 3; python scripts/dev/cont_probe.py 6; python scripts/dev/cont_probe.py 10 ) 2>/dev/null | grep -v amdgpu > $O/continuation_pass.txt
ls $O
