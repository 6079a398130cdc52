#!/bin/bash
# round 4, GPU session D: i8 operand prefetch A/B, the trajectory kernel's new arithmetic (tests + c5), PMC of AUTO
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( python tools/bench_i8.py i8x4 i8x5 auto
  TGP_LIB=$PWD/tools/exp/libtgp_pf0.so python tools/bench_i8.py i8x4 i8x5 auto ) 2>&1 | grep -v amdgpu.ids | tee $OUT/r4d_i8_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_host.py tests/test_gpu_i8.py -q -x -k "golden or traj or thompson or c5 or rff or i8x4_sweep or auto_precision_stays or gibbon or entropy" 2>&1 | tail -8 | tee $OUT/r4d_tests.txt
timeout 200 python bench.py --workload c5 --no-cpu-baseline --no-acquire --no-secondary --steps 5 > $OUT/r4d_bench_c5.json 2> $OUT/r4d_bench_c5.err; python -c "
import json; j=json.load(open('gpurun_out/r4d_bench_c5.json')); print('c5', j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'])"
timeout 600 tools/gpu_profile.sh r04 auto
cat $OUT/prof_r04_auto_summary.log | tail -3
