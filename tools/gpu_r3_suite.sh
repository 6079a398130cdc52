#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
