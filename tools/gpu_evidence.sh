#!/bin/bash
# One GPU-box session (via gpurun, from the repo root): the closing evidence of a round with the FINAL library.
#   usage: tools/gpu_evidence.sh <round> [bench|traffic|update|fit|all]
#   bench   : the driver's command (python bench.py --steps 20 --warmup 5) plain and under rocprofv3 --kernel-trace --stats
#   traffic : FETCH_SIZE / WRITE_SIZE passes (separate rocprofv3 --pmc runs, --kernel-trace only) of every bench line's
#             dominant kernel -> profiles/traffic.json entries labelled with this round; wait-state counters of the AUTO sweep
#   update  : `update` at N = 4096 / 8192: kernel trace, the persistent kernel's own time stamps, call latency
#   fit     : the BO-step timings of ONE session (find_best_model_initialization, cold optimize, acquire)
set -u; R=${1:-r05}; WHAT=${2:-all}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
want() { [ "$WHAT" = all ] || [ "$WHAT" = "$1" ]; }
if want bench; then
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_default.json 2> $OUT/${R}_bench_default.err; echo "bench rc=$?"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R}_default_stats -o stats -- python $OLDPWD/bench.py > $OUT/${R}_bench_default_under_rocprof.json 2> $OUT/prof_${R}_default.log ); echo "default stats rc=$?"
  python - $R <<'PY' > $OUT/${R}_rocprof_default_command.txt 2>&1
import sqlite3, glob, sys
R = sys.argv[1]
p = glob.glob(f'gpurun_out/prof_{R}_default_stats/*_results.db')[0]
cur = sqlite3.connect(p).cursor()
print("rocprofv3 --kernel-trace --stats -- python bench.py   (the default command: headline + secondary workloads + cpu_baseline)")
print(f"{'kernel':100s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>7s}")
for name, calls, total, avg, pct in cur.execute("select * from top_kernels limit 30"):
    print(f"{name[:100]:100s} {calls:6d} {total:14.0f} {avg:14.0f} {pct:7.2f}")
PY
  rm -rf $OUT/prof_${R}_default_stats
  python - $R <<'PY'
import json, sys
j = json.load(open(f'gpurun_out/{sys.argv[1]}_bench_default.json'))
print('headline', j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'], 'traffic', j['roofline'].get('traffic'), 'update_ms', j['config']['update_ms'], 'fit', j['config'].get('fit'), 'acquire', j['config'].get('acquire_ms'))
for k, v in j.get('secondary', {}).items():
    r = v.get('roofline', {})
    print(k, v.get('value'), r.get('frac'), r.get('kernel_ms'), 'traffic', r.get('traffic'), v.get('auto'), v.get('error'))
print(j['cpu_baseline']['value'], j['cpu_baseline']['cores'])
PY
  head -9 $OUT/${R}_rocprof_default_command.txt
fi
if want traffic; then
  for W in headline c2 c4 c5 auto i8x5; do
    WL=$W; EXTRA=""
    [ "$W" = auto ] && { WL=headline; EXTRA="--precision auto"; }
    [ "$W" = i8x5 ] && { WL=headline; EXTRA="--precision i8x5"; }
    B="python $PWD/bench.py --workload $WL $EXTRA --no-cpu-baseline --no-acquire --no-secondary --steps 1 --warmup 0"
    ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_${R}_${W}_fetch -o fetch -- $B > $OUT/prof_${R}_${W}_fetch.log 2>&1 ); echo "$W fetch rc=$?"
    ( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_${R}_${W}_write -o write -- $B > $OUT/prof_${R}_${W}_write.log 2>&1 ); echo "$W write rc=$?"
    if [ "$W" = auto ] || [ "$W" = c5 ]; then
      ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/prof_${R}_${W}_wait -o wait -- $B > $OUT/prof_${R}_${W}_wait.log 2>&1 ); echo "$W wait rc=$?"
    fi
    SUMMARY_DIR=$OUT python tools/summarize_rocprof.py $R $W > $OUT/prof_${R}_${W}_summary.log 2>&1; echo "$W summary rc=$?"
    rm -rf $OUT/prof_${R}_${W}_fetch $OUT/prof_${R}_${W}_write $OUT/prof_${R}_${W}_wait
  done
  python -c "import json; t=json.load(open('gpurun_out/traffic.json')); print({k:v for k,v in t.items() if not k.startswith('_')}, t.get('_rounds'))"
fi
if want update; then
  {
  echo "# One update at N = 4096 / 8192 with the final library"
  echo "# (a) rocprofv3 --kernel-trace of the last set_data (tools/gpu_upd_trace.sh): launches per update"
  for N in 4096 8192; do echo "## N = $N"; timeout 200 tools/gpu_upd_trace.sh $N 2>&1 | grep -v "^rc="; done
  echo
  echo "# (b) the persistent kernel's own time stamps (TGP_DAG_TRACE, tools/dag_trace.py; the stamps cost ~3 %)"
  for N in 4096 8192; do echo "## N = $N"; TGP_DAG_TRACE=/tmp/dag_trace_$N.bin timeout 200 python tools/dag_trace.py $N 2>&1 | grep -v amdgpu.ids; done
  } > $OUT/${R}_update_breakdown.txt 2>&1
  head -40 $OUT/${R}_update_breakdown.txt
  timeout 100 python tools/bench_update.py 4096 8192 2>&1 | grep -v amdgpu.ids | tee $OUT/${R}_update_latency.txt
fi
if want fit; then
  { timeout 300 python tools/bench_bo_step.py 4096 2>&1 | grep -v amdgpu.ids; timeout 100 python tools/bench_cold_fit.py 4096 2>&1 | grep -v amdgpu.ids; } | tee $OUT/${R}_bo_step.txt
fi
