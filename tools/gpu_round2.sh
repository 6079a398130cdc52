#!/bin/bash
# One GPU-box session of round 2: new tests first, the whole GPU suite, bench lines of every workload, profiles.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
STAGE=${1:-all}
if [ "$STAGE" = all ] || [ "$STAGE" = tests ]; then
  timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_philox.py tests/test_gpu_c3.py -m gpu -x -q > $OUT/r02_new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 $OUT/r02_new_tests.log
  timeout 900 python -m pytest tests -m gpu -q > $OUT/r02_pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -8 $OUT/r02_pytest_gpu.log
fi
if [ "$STAGE" = all ] || [ "$STAGE" = bench ]; then
  for W in headline c2 c4 c5; do
    timeout 600 python bench.py --workload $W > $OUT/r02_bench_$W.json 2> $OUT/r02_bench_$W.err; echo "bench $W rc=$?"; cat $OUT/r02_bench_$W.json; tail -3 $OUT/r02_bench_$W.err
  done
  timeout 300 python bench.py --gpus 1 --mode group --workload headline --no-cpu-baseline --no-acquire > $OUT/r02_bench_group.json 2> $OUT/r02_bench_group.err; echo "group rc=$?"; cat $OUT/r02_bench_group.json
fi
if [ "$STAGE" = all ] || [ "$STAGE" = prof ]; then
  for W in c4 c5 c2; do timeout 900 tools/gpu_profile.sh r02 $W; done
fi
