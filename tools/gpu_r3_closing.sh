#!/bin/bash
# closing pass: full GPU suite (margins table), then the evidence run of tools/gpu_r3_evidence.sh
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 900 tools/gpu_r3_evidence.sh
