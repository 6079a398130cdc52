#!/bin/bash
# GPU batch (round 6): pipelined sub, asynchronous flag look
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dag.py -x -q 2>&1 | tail -3
{
for lib in tools/exp/libtgp_dagpipe0.so trieste_amd/libtgp.so; do echo "== $lib"; TGP_LIB=$PWD/$lib timeout 200 python tools/bench_update.py 4096 2>&1 | grep -v amdgpu.ids; done
echo "== trace, pipelined"; TGP_DAG_TRACE=/tmp/dag_trace_p.bin timeout 200 python tools/dag_trace.py 4096 2>&1 | grep -v amdgpu.ids | head -8
} 2>&1 | tee $OUT/r06_dag_pipe2.txt
