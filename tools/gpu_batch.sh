#!/bin/bash
# GPU batch (round 6, closing): the whole suite and the default bench line with the final library
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_suite.sh r06d 2>&1 | tail -6
bash tools/gpu_evidence.sh r06d bench 2>&1 | tail -20
