#!/bin/bash
# GPU batch (round 6, closing): kernel trace of the qEI value-and-gradient call and of the small joint calls
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_qeigrad -o q -- python $OLDPWD/tools/bench_qei_grad.py 4096 5 > $OUT/prof_qeigrad.log 2>&1 ); echo "rc=$?"
python - <<'PY' > $OUT/r06_qei_grad_trace.txt 2>&1
import sqlite3, glob
p = glob.glob('gpurun_out/prof_qeigrad/*_results.db')[0]
cur = sqlite3.connect(p).cursor()
print("rocprofv3 --kernel-trace --stats -- python tools/bench_qei_grad.py 4096 5   (N = 4096, groups of 5: joint_forward / predict_joint / joint_vjp / qei_value_grad at 10, 60, 300 groups, then two EGO acquires)")
print(f"{'kernel':110s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
for name, calls, total, avg, pct in cur.execute("select * from top_kernels limit 28"):
    print(f"{name[:110]:110s} {calls:6d} {total:12.0f} {avg:10.1f} {pct:6.2f}")
PY
grep -v amdgpu $OUT/prof_qeigrad.log | tail -6 | cut -c1-300 >> $OUT/r06_qei_grad_trace.txt
rm -rf $OUT/prof_qeigrad
head -34 $OUT/r06_qei_grad_trace.txt | cut -c1-160
