#!/bin/bash
# GPU batch (round 6): the whole suite with tgp_predict at <= 2048 points as a skinny product
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_suite.sh r06h 2>&1 | tail -6
