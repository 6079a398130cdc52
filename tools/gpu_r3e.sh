#!/bin/bash
# round 3: full GPU suite (margins table), update latency / trace, bench with secondary lines
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
echo "== update latency"; timeout 120 python tools/bench_update.py 1024 2048 4096 8192 2>&1 | grep update
echo "== update trace"; timeout 200 bash tools/gpu_upd_trace.sh 4096 2>&1 | tail -12
echo "== bench"; timeout 600 python bench.py > $OUT/bench_r3e.json 2> $OUT/bench_r3e.err; tail -c 1500 $OUT/bench_r3e.json; tail -3 $OUT/bench_r3e.err
