#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "qei or reparam or joint or mc_ei or batch" 2>&1 | tail -5
timeout 300 python bench.py --workload c4 --steps 3 --no-cpu-baseline --no-secondary --no-acquire > $OUT/bench_r3i_c4.json 2> $OUT/bench_r3i_c4.err; echo "c4 rc=$?"; cat $OUT/bench_r3i_c4.json | cut -c1-900
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c4 -o stats -- python $OLDPWD/bench.py --workload c4 --steps 2 --no-cpu-baseline --no-secondary --no-acquire > /dev/null 2>&1 )
python - <<'PY'
import sqlite3, glob
p = glob.glob('gpurun_out/prof_c4/*_results.db')[0]
cur = sqlite3.connect(p).cursor()
for name, calls, total, avg, pct in cur.execute("select * from top_kernels limit 6"):
    print(f"{name[:90]:90s} {calls:6d} {total:14.0f} {avg:14.0f} {pct:7.2f}")
PY
rm -rf $OUT/prof_c4
