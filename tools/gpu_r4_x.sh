#!/bin/bash
# round 4, the last seconds: FETCH_SIZE of the rebuilt joint kernel at c4 (one PMC pass; WRITE_SIZE is owed)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 38 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_r04_c4_fetch -o fetch -- python $OLDPWD/bench.py --workload c4 --no-cpu-baseline --no-acquire --no-secondary --steps 1 --warmup 0 > $OUT/prof_r04_c4_fetch.log 2>&1 ); echo "rc=$?"
python - <<'PY' 2>&1 | tee gpurun_out/r04_c4_fetch.txt
import sqlite3, glob
p = glob.glob('gpurun_out/prof_r04_c4_fetch/*_results.db')[0]
cur = sqlite3.connect(p).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if 'pmc_event' in t][0]; info = [t for t in tabs if 'info_pmc' in t][0]
disp = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, d.end - d.start, sum(e.value) from {pmc} e join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id group by d.id order by d.start"
for name, dur, val in cur.execute(q):
    if 'joint_kernel' in name:
        print(f"joint_kernel dispatch: {dur/1e6:.1f} ms  FETCH_SIZE sum = {val:.6g} (KiB)  -> 2 x FETCH = {2*val*1024/1e9:.1f} GB")
PY
rm -rf $OUT/prof_r04_c4_fetch
