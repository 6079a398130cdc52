#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python bench.py --no-secondary --no-cpu-baseline --steps 1 --warmup 0 > $OUT/bench_fit_check.json 2> $OUT/bench_fit_check.err; echo "rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/bench_fit_check.json').read().strip().splitlines()[-1]);print(d['config'].get('fit'), d['config'].get('acquire_ms'), d['value'])"
