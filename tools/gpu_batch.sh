#!/bin/bash
# GPU batch (round 6, closing): the whole suite, the default bench line, the fit and update evidence with the final library (persistent update kernel from Npad = 512 on)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_suite.sh r06c 2>&1 | tail -6
bash tools/gpu_evidence.sh r06c bench 2>&1 | tail -20
bash tools/gpu_evidence.sh r06c fit 2>&1 | tail -12
bash tools/gpu_evidence.sh r06c update 2>&1 | tail -4
