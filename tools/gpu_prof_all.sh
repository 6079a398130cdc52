#!/bin/bash
set -u
for W in "$@"; do timeout 600 tools/gpu_profile.sh r02 $W; done
ls -la gpurun_out | head -40
