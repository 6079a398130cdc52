#!/bin/bash
# round 4, GPU session A: the whole -m gpu suite (new qEI / AUTO tests), then the default bench line
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 -s 2>&1 | grep -v "^$" | tail -60 > $OUT/r4a_tests.txt; tail -25 $OUT/r4a_tests.txt
cp gpurun_out/parity_margins.txt $OUT/r4a_parity_margins.txt 2>/dev/null
timeout 500 python bench.py > $OUT/r4a_bench.json 2> $OUT/r4a_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r4a_bench.json'))
print('headline', j['value'], j['roofline']['frac'], 'update_ms', j['config']['update_ms'], 'fit', j['config'].get('fit'))
for k,v in j.get('secondary',{}).items():
    print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('kernel_ms'), v.get('auto'), v.get('error'))
PY
