#!/bin/bash
# GPU batch (round 6): C4 with the joint kernel back at its round-5 source, against the round-5 tree, one box
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { timeout 300 python bench.py --workload c4 --no-cpu-baseline --no-acquire --no-secondary --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
{ for rep in 1 2; do echo "== r06 tree (joint kernel = round-5 source)"; run; echo "== r05 tree"; ( cd tools/exp/r05tree && run ); done; } | tee $OUT/r06_joint_reverted_c4.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
