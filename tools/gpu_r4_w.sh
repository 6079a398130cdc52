#!/bin/bash
# round 4: rocprofv3 kernel stats of the headline workload with the final kernel (the default-command stats file is one commit older)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 45 rocprofv3 --kernel-trace --stats -d $OUT/prof_r04_final -o stats -- python $OLDPWD/bench.py --workload headline --no-cpu-baseline --no-acquire --no-secondary --steps 4 --warmup 1 > $OUT/r04_headline_under_rocprof.json 2> $OUT/prof_r04_final.log ); echo "rc=$?"
python - <<'PY' > gpurun_out/r04_rocprof_headline_final.txt 2>&1
import sqlite3, glob, json
p = glob.glob('gpurun_out/prof_r04_final/*_results.db')[0]
cur = sqlite3.connect(p).cursor()
print("rocprofv3 --kernel-trace --stats -- python bench.py --workload headline --no-secondary --steps 4 --warmup 1   (final headline kernel)")
print(f"{'kernel':100s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>7s}")
for name, calls, total, avg, pct in cur.execute("select * from top_kernels limit 6"):
    print(f"{name[:100]:100s} {calls:6d} {total:14.0f} {avg:14.0f} {pct:7.2f}")
j = json.load(open('gpurun_out/r04_headline_under_rocprof.json'))
print("bench line of the same run: kernel_ms (HIP events) =", j['roofline']['kernel_ms'], " frac =", j['roofline']['frac'])
PY
rm -rf $OUT/prof_r04_final; cat gpurun_out/r04_rocprof_headline_final.txt | cut -c1-160
