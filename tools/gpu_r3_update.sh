#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for N in 4096 8192; do
  TGP_DAG_TRACE=/tmp/tr_$N.bin timeout 120 python tools/dag_trace.py $N > $OUT/dag_trace_inorder_$N.txt 2>&1; echo "trace $N rc=$?"
done
head -8 $OUT/dag_trace_inorder_4096.txt; head -4 $OUT/dag_trace_inorder_8192.txt
timeout 200 python tools/bench_update.py 2>&1 | tail -12
timeout 300 python -m pytest tests/test_gpu_dag.py -q -x 2>&1 | tail -3
