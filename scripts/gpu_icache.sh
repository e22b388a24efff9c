#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ic; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_BUSY_CY[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*" | sort -u | tr '\n' ' '
echo
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --check 0"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d gpurun_out/ic/a -o pmc -- $CMD > gpurun_out/ic/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/ic/b -o pmc -- $CMD > gpurun_out/ic/b.log 2>&1
python - <<'PY'
import csv, collections, glob
for d in ("a","b"):
    for f in glob.glob(f"gpurun_out/ic/{d}/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(list)
        for r in rows:
            if "hmpc_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: sum(v)/len(v) for k,v in agg.items()})
PY
tail -3 gpurun_out/ic/a.log
