#!/bin/bash
# kernel trace of ONE update at N (default 4096): per (kernel, grid) durations of the last set_data
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; N=${1:-4096}
( cd /tmp && rocprofv3 --kernel-trace -d $OUT/prof_upd_trace -o upd -- python $OLDPWD/tools/prof_update.py $N > $OUT/prof_upd_trace.log 2>&1 ); echo rc=$?
python - <<'PY'
import sqlite3, glob, collections
p=glob.glob('gpurun_out/prof_upd_trace/*_results.db')[0]
cur=sqlite3.connect(p).cursor()
rows=list(cur.execute("select name, start, end, grid_x/workgroup_x from kernels order by start"))
idx=[i for i,r in enumerate(rows) if 'assemble_K' in r[0]]
seg=rows[idx[-1]:]
e=[i for i,r in enumerate(seg) if 'trmv' in r[0]]
seg=seg[:e[-1]+1] if e else seg
print(f"update span {1e-3*(seg[-1][2]-seg[0][1]):.1f} us, {len(seg)} kernels")
agg=collections.OrderedDict()
for r in seg:
    k=(r[0].split('(')[0].replace('void ','').replace('tgp::','')[:34], r[3])
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=r[2]-r[1]
print("kernels with 256 workgroups, in order (us):", " ".join(f"{1e-3*(r[2]-r[1]):.0f}" for r in seg if r[3]==256))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print(f"{k[0]:34s} wgs={k[1]:5d} n={v[0]:4d} avg={1e-3*v[1]/v[0]:7.1f} us  total={1e-3*v[1]:8.1f} us")
PY
rm -rf $OUT/prof_upd_trace
