#!/bin/bash
# round 4: headline kernel with the generated K* rows ahead of the step's DMA (A/B), c4 through bench.py with the rebuilt joint kernel
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in dgen0 dgen1 dgen0 dgen1; do
  echo "== $v: $(TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 120 python bench.py --workload headline --no-cpu-baseline --no-acquire --no-secondary --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'])")"
done | tee $OUT/r04_dma_genfirst.txt
for v in dgen0 dgen1; do
  echo "== c2 $v: $(TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 120 python bench.py --workload c2 --no-cpu-baseline --no-acquire --no-secondary --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'])")"
done | tee -a $OUT/r04_dma_genfirst.txt
echo "== c4 (rebuilt joint kernel): $(timeout 200 python bench.py --workload c4 --no-cpu-baseline --no-acquire --no-secondary --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'])")" | tee -a $OUT/r04_dma_genfirst.txt
