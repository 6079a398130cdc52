set -u
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_sweep_matches_oracle or launch_policies or joint or qei" > gpurun_out/r02_stage_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02_stage_tests.log
run() { TGP_LIB=$1 timeout 200 python bench.py --workload $3 $4 --steps 3 --no-cpu-baseline --no-acquire 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 $3', 'value', o['value'], 'kernel_ms', round(o['roofline']['kernel_ms'],2), 'frac', round(o['roofline']['frac'],4))"; }
for W in headline c2 c4; do
run "" new $W ""
run $PWD/tools/exp/libtgp_x32.so old $W ""
done
