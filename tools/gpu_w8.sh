set -u; OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_sweep_matches_oracle" > $OUT/r02_w8_tests.log 2>&1; echo "w8 tests rc=$?"; tail -5 $OUT/r02_w8_tests.log
for V in 0 4; do for W in headline c2; do
timeout 300 python bench.py --workload $W --variant $V --no-cpu-baseline --no-acquire --steps 3 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $V $W', o['value'], o['roofline']['frac'], o['roofline']['kernel_ms'])"
done; done
