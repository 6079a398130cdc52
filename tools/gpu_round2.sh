#!/bin/bash
# One GPU-box session of round 2: the whole GPU suite, bench lines of every workload, profiles.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
STAGE=${1:-all}
if [ "$STAGE" = all ] || [ "$STAGE" = tests ]; then
  timeout 1200 python -m pytest tests -m gpu -q > $OUT/r02_pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -8 $OUT/r02_pytest_gpu.log
fi
if [ "$STAGE" = all ] || [ "$STAGE" = bench ]; then
  for W in headline c2 c3 c4 c5; do
    timeout 600 python bench.py --workload $W > $OUT/r02_bench_$W.json 2> $OUT/r02_bench_$W.err; echo "bench $W rc=$?"; cat $OUT/r02_bench_$W.json; tail -2 $OUT/r02_bench_$W.err
  done
  timeout 300 python bench.py --precision i8x4 > $OUT/r02_bench_i8.json 2> $OUT/r02_bench_i8.err; echo "bench i8 rc=$?"; cat $OUT/r02_bench_i8.json
  timeout 300 python bench.py --precision i8x5 --no-cpu-baseline > $OUT/r02_bench_i8x5.json 2> $OUT/r02_bench_i8x5.err; echo "bench i8x5 rc=$?"; cat $OUT/r02_bench_i8x5.json
  timeout 300 python bench.py --gpus 1 --mode group --workload headline --no-cpu-baseline --no-acquire > $OUT/r02_bench_group.json 2> $OUT/r02_bench_group.err; echo "group rc=$?"; cat $OUT/r02_bench_group.json
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --workload c3 --no-cpu-baseline --no-acquire > $OUT/r02_bench_c3_torchrun1.json 2> $OUT/r02_bench_c3_torchrun1.err; echo "torchrun rc=$?"; cat $OUT/r02_bench_c3_torchrun1.json
  python tools/bench_update.py 1024 2048 4096 8192 > $OUT/r02_update.txt 2>&1; cat $OUT/r02_update.txt
fi
if [ "$STAGE" = all ] || [ "$STAGE" = prof ]; then
  for W in headline i8 c4 c5 c2; do timeout 900 tools/gpu_profile.sh r02 $W; done
fi
