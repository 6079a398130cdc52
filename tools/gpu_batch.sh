#!/bin/bash
# GPU batch (round 6, closing): the --runslow twins with the final library
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --runslow only 2>&1 | tail -6 | tee $OUT/r06_gpu_tests_runslow.txt
