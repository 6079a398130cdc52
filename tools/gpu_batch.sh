#!/bin/bash
# GPU batch (round 6, closing): the whole suite with the final library and tests
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_suite.sh r06m 2>&1 | tail -14
