#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "traj or thompson or c5 or gibbon or kernel_sums or rff" 2>&1 | tail -4
timeout 300 python bench.py --workload c5 --steps 5 --no-cpu-baseline --no-secondary --no-acquire > $OUT/bench_r3n_c5.json 2> $OUT/bench_r3n_c5.err; echo "c5 rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r3n_c5.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])
PY
