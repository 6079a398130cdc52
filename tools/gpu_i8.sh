set -u; OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_i8.py tests/test_philox.py -m gpu -q -s > $OUT/r02_i8_tests.log 2>&1; echo "i8 tests rc=$?"; tail -25 $OUT/r02_i8_tests.log
timeout 300 python bench.py --precision i8x4 --no-cpu-baseline --no-acquire > $OUT/r02_bench_i8.json 2> $OUT/r02_bench_i8.err; echo "bench i8 rc=$?"; cat $OUT/r02_bench_i8.json; tail -3 $OUT/r02_bench_i8.err
