#!/bin/bash
# closing pass of round 3: full GPU suite (margins table), the evidence run, smoke, the kernel stats of the default command
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r03_smoke.log 2>&1; echo "smoke rc=$? ($(grep -c 'smoke ok' $OUT/r03_smoke.log))"
timeout 900 tools/gpu_r3_evidence.sh
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
timeout 200 python tools/bench_bo_step.py 4096 2>&1 | grep -v amdgpu.ids > $OUT/r03_bo_step.txt; cat $OUT/r03_bo_step.txt
