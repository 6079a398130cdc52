#!/bin/bash
# GPU batch (round 6, closing): the driver's command twice more with the final library (plain)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_default_final_$i.json 2> /dev/null; echo "rc=$?"; done
python - <<'PY'
import json
for i in (1, 2):
    j = json.load(open(f'gpurun_out/r06_bench_default_final_{i}.json'))
    print(i, j['value'], j['roofline']['frac'], j['config']['update_ms'], j['config']['acquire_ms'], j['config']['fit']['ms'], j['config']['fit']['second_ms'],
          {k: v.get('value') for k, v in j['secondary'].items()})
PY
