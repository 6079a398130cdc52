#!/bin/bash
# round 4, last session: the sweep parity tests and the driver's bench command with the final headline kernel
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 110 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not c4 and not c5 and not qei and not joint" 2>&1 | tail -4 | tee $OUT/r04_final_sweep_tests.txt
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_final.json 2> $OUT/r04_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04_bench_final.json'))
print('headline', j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'])
for k,v in j.get('secondary',{}).items():
    print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('kernel_ms'), v.get('same_winner_as_f64_headline'))
PY
