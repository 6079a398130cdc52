set -u; OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_i8.py tests/test_philox.py -m gpu -q -s > $OUT/r02_i8_tests.log 2>&1; echo "i8 tests rc=$?"; tail -25 $OUT/r02_i8_tests.log
for P in i8x4 i8x5; do
timeout 300 python bench.py --precision $P --no-cpu-baseline --no-acquire --steps 3 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P', o['value'], o['roofline']['frac'], o['roofline']['kernel_ms'], o['config']['best_index'], o['config']['best_value'])"
done
