#!/bin/bash
# round 4, last seconds of budget: the int8 sweep with its second row fragment deferred across the barrier (TGP_I8_DEFER,
# off by default) against the shipped kernel -- timing and winner only, NOT a validation
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in "" i8d; do
  L=$PWD/trieste_amd/libtgp.so; [ -n "$v" ] && L=$PWD/tools/exp/libtgp_$v.so
  echo "== auto ${v:-shipped}: $(TGP_LIB=$L timeout 40 python bench.py --workload headline --precision auto --no-cpu-baseline --no-acquire --no-secondary --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline'].get('kernel_ms'), j.get('best'), j.get('auto') or j['config'].get('auto'))")"
done | tee $OUT/r04_i8_defer.txt
