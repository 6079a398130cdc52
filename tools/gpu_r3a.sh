#!/bin/bash
# round 3, GPU session A: the persistent update kernel first (correctness, then latency), then the new tests, then bench
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== dag tests"; timeout 900 python -m pytest tests/test_gpu_dag.py -m gpu -x -q 2>&1 | tail -15
echo "== update latency (DAG)"; timeout 300 python tools/bench_update.py 512 1024 2048 4096 8192 2>&1 | grep -v Warning | tail -6
echo "== update latency (recursion)"; TGP_NO_DAG=1 timeout 300 python tools/bench_update.py 512 1024 2048 4096 8192 2>&1 | grep -v Warning | tail -6
echo "== update trace"; timeout 300 bash tools/gpu_upd_trace.sh 4096 2>&1 | tail -14
echo "== i8 / auto tests"; timeout 900 python -m pytest tests/test_gpu_i8.py -m gpu -x -q -s 2>&1 | grep -E "margin|passed|failed|Error|error" | tail -40
echo "== bench"; timeout 900 python bench.py > $OUT/bench_r3a.json 2> $OUT/bench_r3a.err; tail -c 3000 $OUT/bench_r3a.json; tail -5 $OUT/bench_r3a.err
