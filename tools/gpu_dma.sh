set -u
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sweep_matches or launch_policies or ties" > gpurun_out/r02_dma_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r02_dma_tests.log
run() { timeout 300 python bench.py --workload $1 --variant $2 --steps 3 --no-cpu-baseline --no-acquire 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 variant $2', 'value', o['value'], 'kernel_ms', round(o['roofline']['kernel_ms'],2), 'frac', round(o['roofline']['frac'],4), o['config']['best_index'], o['config']['best_value'])"; }
for W in headline c2 c3; do run $W 0; done
