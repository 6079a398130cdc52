set -u
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "joint or qei or c4" > gpurun_out/r02_joint_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r02_joint_tests.log
for V in 0 4; do
timeout 300 python bench.py --workload c4 --variant $V --steps 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $V c4 value', o['value'], 'ms/step', round(o['ms_per_step'],1), 'kernel_ms', round(o['roofline']['kernel_ms'],1), 'frac', round(o['roofline']['frac'],4))"
done
