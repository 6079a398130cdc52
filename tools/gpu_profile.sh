#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel stats + separate PMC passes for one
# bench workload.   usage: tools/gpu_profile.sh <round> <workload> [extra bench args]
# (counters are collected in their own runs with --kernel-trace only, as the pool requires)
set -u
ROUND=${1:-r03}; W=${2:-headline}; shift 2 || true
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
WL=$W; EXTRA=""
if [ "$W" = i8 ]; then WL=headline; EXTRA="--precision i8x4"; fi   # the split-precision line of the headline workload
if [ "$W" = i8x5 ]; then WL=headline; EXTRA="--precision i8x5"; fi
if [ "$W" = auto ]; then WL=headline; EXTRA="--precision auto"; fi # ... with the a-posteriori float64 repair (TGP_PREC_AUTO)
B="python $PWD/bench.py --workload $WL $EXTRA --no-cpu-baseline --no-acquire --no-secondary $*"
T=${ROUND}_${W}
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_${T}_stats -o stats -- $B --steps 3 --warmup 1 > $OUT/prof_${T}_stats.log 2>&1 ); echo "$W stats rc=$?"
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_${T}_fetch -o fetch -- $B --steps 1 --warmup 0 > $OUT/prof_${T}_fetch.log 2>&1 ); echo "$W fetch rc=$?"
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_${T}_write -o write -- $B --steps 1 --warmup 0 > $OUT/prof_${T}_write.log 2>&1 ); echo "$W write rc=$?"
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/prof_${T}_sq -o sq -- $B --steps 1 --warmup 0 > $OUT/prof_${T}_sq.log 2>&1 ); echo "$W sq rc=$?"
( cd /tmp && rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/prof_${T}_wait -o wait -- $B --steps 1 --warmup 0 > $OUT/prof_${T}_wait.log 2>&1 ); echo "$W wait rc=$?"
# summarise on the box and drop the raw databases (gpurun_out/ travels back only below 64 MiB)
SUMMARY_DIR=$OUT python tools/summarize_rocprof.py $ROUND $W > $OUT/prof_${T}_summary.log 2>&1; echo "$W summary rc=$?"
rm -rf $OUT/prof_${T}_stats $OUT/prof_${T}_fetch $OUT/prof_${T}_write $OUT/prof_${T}_sq $OUT/prof_${T}_wait
