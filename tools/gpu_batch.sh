#!/bin/bash
# GPU batch (round 6, closing): the default bench line with the final library
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_evidence.sh r06f bench 2>&1 | tail -20
