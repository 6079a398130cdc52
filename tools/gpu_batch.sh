#!/bin/bash
# GPU batch (round 6): the -m gpu suite including its --runslow twins, with the final library
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu --runslow --durations=8 2>&1 | grep -v "^$" | tail -16 > $OUT/r06_gpu_tests_runslow.txt; cat $OUT/r06_gpu_tests_runslow.txt
