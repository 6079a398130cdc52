#!/bin/bash
# GPU batch (round 6, closing): the whole suite with the final library
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_suite.sh r06g 2>&1 | tail -5
