#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
echo "# One update at N = 4096 / 8192, round 4 HEAD (tile task: staggered operand requests + s_setprio; chain unchanged)"
echo "# (a) rocprofv3 --kernel-trace of the last set_data (tools/gpu_upd_trace.sh): launches per update"
for N in 4096 8192; do echo "## N = $N"; timeout 200 tools/gpu_upd_trace.sh $N 2>&1 | grep -v "^rc="; done
echo
echo "# (b) the persistent kernel's own time stamps (TGP_DAG_TRACE, tools/dag_trace.py; the stamps cost ~3 %)"
for N in 4096 8192; do echo "## N = $N"; TGP_DAG_TRACE=/tmp/dag_trace_$N.bin timeout 200 python tools/dag_trace.py $N 2>&1 | grep -v amdgpu.ids; done
} > $OUT/r04_update_breakdown.txt 2>&1
cat $OUT/r04_update_breakdown.txt | head -60
