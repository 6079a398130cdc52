#!/bin/bash
# One GPU-box session (via gpurun, from the repo root): the whole `-m gpu` suite, smoke(), the parity-margin table.
#   usage: tools/gpu_suite.sh <round>      -> gpurun_out/<round>_gpu_tests.txt, <round>_parity_margins.txt
set -u; R=${1:-r05}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | grep -v "^$" | tail -16 > $OUT/${R}_gpu_tests.txt; cat $OUT/${R}_gpu_tests.txt
cp gpurun_out/parity_margins.txt $OUT/${R}_parity_margins.txt 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke\|error\|Traceback" | tail -3 | tee -a $OUT/${R}_gpu_tests.txt
