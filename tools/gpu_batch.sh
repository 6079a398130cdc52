#!/bin/bash
# GPU batch (round 6, closing): the whole suite and the evidence of `update` / the default bench line / the fit with the final library
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_suite.sh r06b 2>&1 | tail -6
bash tools/gpu_evidence.sh r06b update 2>&1 | tail -5
bash tools/gpu_evidence.sh r06b bench 2>&1 | tail -20
bash tools/gpu_evidence.sh r06b fit 2>&1 | tail -12
