#!/bin/bash
# round-3 evidence: rocprofv3 stats + PMC passes of the headline workload, the stats of the DEFAULT bench command,
# kernel trace of one update at N = 4096 and 8192
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 tools/gpu_profile.sh r03 headline
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03_default_stats -o stats -- python $OLDPWD/bench.py > $OUT/r03_bench_default_under_rocprof.json 2> $OUT/prof_r03_default.log ); echo "default stats rc=$?"
python - <<'PY' > gpurun_out/r03_rocprof_default_command.txt 2>&1
import sqlite3, glob
p = glob.glob('gpurun_out/prof_r03_default_stats/*_results.db')[0]
cur = sqlite3.connect(p).cursor()
print("rocprofv3 --kernel-trace --stats -- python bench.py   (the default command: headline + secondary workloads + cpu_baseline)")
print(f"{'kernel':100s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>7s}")
for name, calls, total, avg, pct in cur.execute("select * from top_kernels limit 25"):
    print(f"{name[:100]:100s} {calls:6d} {total:14.0f} {avg:14.0f} {pct:7.2f}")
PY
rm -rf $OUT/prof_r03_default_stats
for N in 4096 8192; do timeout 200 tools/gpu_upd_trace.sh $N > $OUT/r03_update_kernels_$N.txt 2>&1; done
cat $OUT/r03_update_kernels_4096.txt | head -20
tail -5 $OUT/r03_rocprof_default_command.txt
