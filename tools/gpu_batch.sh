#!/bin/bash
# GPU batch (round 6, closing): the driver's command with the small-call timings in config
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_default_final_3.json 2> $OUT/r06_bench_default_final_3.err; echo "rc=$?"
python - <<'PY'
import json
j = json.load(open('gpurun_out/r06_bench_default_final_3.json'))
print(j['value'], j['roofline']['frac'], j['config']['update_ms'], j['config']['acquire_ms'], j['config']['fit'], j['config']['small_calls_ms'])
print({k: v.get('value') for k, v in j['secondary'].items()})
PY
timeout 300 python -m pytest tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -3
