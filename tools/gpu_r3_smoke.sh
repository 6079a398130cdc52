#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 400 python bench.py > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err; echo "bench rc=$?"
