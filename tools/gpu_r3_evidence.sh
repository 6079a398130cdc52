#!/bin/bash
# round-3 closing evidence: the default bench line, update latency table, update breakdown (kernel's own time stamps),
# kernel trace of one update (number of launches)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python bench.py > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err; echo "bench rc=$?"
timeout 300 python tools/bench_update.py 256 1024 2048 4096 6144 8192 12288 > $OUT/r03_update_latency.txt 2>&1; echo "upd rc=$?"
for N in 4096 8192; do
  TGP_DAG_TRACE=/tmp/tr_$N.bin timeout 120 python tools/dag_trace.py $N > $OUT/dag_trace_inorder_$N.txt 2>&1; echo "trace $N rc=$?"
  timeout 200 bash tools/gpu_upd_trace.sh $N > $OUT/r03_update_kernels_$N.txt 2>&1; echo "ktrace $N rc=$?"
done
cat $OUT/r03_update_latency.txt | grep update; head -14 $OUT/r03_update_kernels_4096.txt
