#!/bin/bash
# GPU batch (round 6): the leaf's worker waves prefetching the rows of P(j+1,j) under the last panel (TGP_DAG_PREFETCH) against the
# shipped form (loads at the product's entry); then the -m gpu suite's --runslow twins
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dag.py -x -q 2>&1 | tail -3 | tee $OUT/r06_dag_prefetch_tests.txt
{
for lib in tools/exp/libtgp_dagpipe0.so trieste_amd/libtgp.so tools/exp/libtgp_dagpipe0.so trieste_amd/libtgp.so; do echo "== $lib"; TGP_LIB=$PWD/$lib timeout 200 python tools/bench_update.py 1024 4096 8192 2>&1 | grep -v amdgpu.ids; done
echo "== trace, prefetch"; TGP_DAG_TRACE=/tmp/dag_trace_p.bin timeout 200 python tools/dag_trace.py 4096 2>&1 | grep -v amdgpu.ids | head -8
} 2>&1 | tee $OUT/r06_dag_prefetch.txt
TGP_LIB=$PWD/tools/exp/libtgp_dagpipe0.so timeout 3000 python -m pytest tests -q -m gpu --runslow only --durations=8 2>&1 | grep -v "^$" | tail -16 > $OUT/r06_gpu_tests_runslow.txt; cat $OUT/r06_gpu_tests_runslow.txt
