#!/bin/bash
# round 4, closing evidence (final code): the whole -m gpu suite, smoke, the driver's bench command, rocprofv3 stats of
# the same command, PMC of c5, the fit timings
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | grep -v "^$" | tail -16 > $OUT/r04_gpu_tests.txt; cat $OUT/r04_gpu_tests.txt
cp gpurun_out/parity_margins.txt $OUT/r04_parity_margins.txt 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_default.json 2> $OUT/r04_bench_default.err; echo "bench rc=$?"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $OUT/prof_r04_default_stats -o stats -- python $OLDPWD/bench.py > $OUT/r04_bench_default_under_rocprof.json 2> $OUT/prof_r04_default.log ); echo "default stats rc=$?"
python - <<'PY' > gpurun_out/r04_rocprof_default_command.txt 2>&1
import sqlite3, glob
p = glob.glob('gpurun_out/prof_r04_default_stats/*_results.db')[0]
cur = sqlite3.connect(p).cursor()
print("rocprofv3 --kernel-trace --stats -- python bench.py   (the default command: headline + secondary workloads + cpu_baseline)")
print(f"{'kernel':100s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>7s}")
for name, calls, total, avg, pct in cur.execute("select * from top_kernels limit 30"):
    print(f"{name[:100]:100s} {calls:6d} {total:14.0f} {avg:14.0f} {pct:7.2f}")
PY
rm -rf $OUT/prof_r04_default_stats
timeout 500 tools/gpu_profile.sh r04 c5
timeout 300 python tools/bench_bo_step.py 4096 2>&1 | grep -v "amdgpu.ids" > $OUT/r04_bo_step.txt
timeout 100 python tools/bench_cold_fit.py 4096 2>&1 | grep -v "amdgpu.ids" >> $OUT/r04_bo_step.txt
timeout 100 python tools/bench_update.py 4096 8192 2>&1 | grep -v "amdgpu.ids" > $OUT/r04_update_latency.txt
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04_bench_default.json'))
print('headline', j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'], 'update_ms', j['config']['update_ms'], 'fit', j['config'].get('fit'), 'acquire', j['config'].get('acquire_ms'))
for k,v in j.get('secondary',{}).items():
    print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('kernel_ms'), v.get('auto'), v.get('error'))
print(j['cpu_baseline']['value'], j['cpu_baseline']['cores'])
PY
head -9 gpurun_out/r04_rocprof_default_command.txt; cat $OUT/r04_bo_step.txt $OUT/r04_update_latency.txt
