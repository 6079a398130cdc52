#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line + rocprofv3 kernel stats + PMC passes.
set -u
ROUND=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 5 --warmup 1 > $OUT/bench_${ROUND}.json 2> $OUT/bench_${ROUND}.err
echo "bench rc=$?"; cat $OUT/bench_${ROUND}.json
B="python $PWD/bench.py --no-cpu-baseline --no-acquire"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_${ROUND}_stats -o stats -- $B --steps 3 --warmup 1 > $OUT/prof_stats.log 2>&1 ); echo "stats rc=$?"
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_${ROUND}_fetch -o fetch -- $B --steps 1 --warmup 0 > $OUT/prof_fetch.log 2>&1 ); echo "fetch rc=$?"
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_${ROUND}_write -o write -- $B --steps 1 --warmup 0 > $OUT/prof_write.log 2>&1 ); echo "write rc=$?"
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/prof_${ROUND}_sq -o sq -- $B --steps 1 --warmup 0 > $OUT/prof_sq.log 2>&1 ); echo "sq rc=$?"
( cd /tmp && rocprofv3 -L > $OUT/rocprof_counters.txt 2>&1 )
find $OUT -name "*.csv" | head -50
# secondary evidence: update / NLML breakdown, other BASELINE configs, GEMM rates
( cd /tmp && R=$OLDPWD && rocprofv3 --kernel-trace --stats -d $OUT/prof_upd -o upd -- python $R/tools/prof_update.py > $OUT/prof_upd.log 2>&1 ); echo "upd rc=$?"
python tools/bench_update.py 1024 2048 4096 8192 > $OUT/update_${ROUND}.txt 2>&1; cat $OUT/update_${ROUND}.txt
python tools/bench_c4c5.py > $OUT/c4c5_${ROUND}.txt 2>&1; cat $OUT/c4c5_${ROUND}.txt
python tools/bench_gemm.py 2048 4096 8192 > $OUT/gemm_${ROUND}.txt 2>&1; tail -15 $OUT/gemm_${ROUND}.txt
python tools/quick_bench.py > $OUT/sizes_${ROUND}.txt 2>&1; tail -12 $OUT/sizes_${ROUND}.txt
# host-level latencies: one BO step, hyper-parameter fit workers, greedy-batch / entropy-search rules
python tools/bench_acquire.py > $OUT/acquire_${ROUND}.txt 2>&1; tail -4 $OUT/acquire_${ROUND}.txt
python tools/bench_optimize.py > $OUT/optimize_${ROUND}.txt 2>&1; tail -3 $OUT/optimize_${ROUND}.txt
python tools/bench_greedy.py 4000 > $OUT/greedy_${ROUND}.txt 2>&1; tail -6 $OUT/greedy_${ROUND}.txt
