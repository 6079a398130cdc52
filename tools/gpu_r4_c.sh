#!/bin/bash
# round 4, GPU session C: whole -m gpu suite, fit timing with the batch breakdown, default bench line
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v "^$" | tail -40 > $OUT/r4c_tests.txt; tail -22 $OUT/r4c_tests.txt
cp gpurun_out/parity_margins.txt $OUT/r4c_parity_margins.txt 2>/dev/null
TGP_TIMING=1 timeout 300 python tools/bench_bo_step.py 4096 2>&1 | grep -v "amdgpu.ids\|chol_inv" | tee $OUT/r4c_bo_step.txt | tail -40
timeout 500 python bench.py > $OUT/r4c_bench.json 2> $OUT/r4c_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r4c_bench.json'))
print('headline', j['value'], j['roofline']['frac'], 'update_ms', j['config']['update_ms'], 'fit', j['config'].get('fit'), 'acquire', j['config'].get('acquire_ms'))
for k,v in j.get('secondary',{}).items():
    print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('kernel_ms'), v.get('auto'), v.get('error'))
print(j['cpu_baseline']['value'], j['cpu_baseline']['cores'])
PY
