#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 8 > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err
tail -3 gpurun_out/bench_r01b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r01b.json').read())
print(d['value'], d['solver'])
for k,v in d['other_configs'].items(): print(k, v)
PY
