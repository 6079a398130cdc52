#!/bin/bash
set -u; export TMPDIR=/tmp
for f in 0; do echo "== fences=$f"; for n in 640 1000 1536 2048; do echo -n "N=$n: "; TGP_DAG_FENCES=$f timeout 300 python tools/dag_debug.py $n 40 2>&1 | grep -v amdgpu.ids | grep -c "NotPositive\|wrong L tiles \[(\|Error" ; done; done
timeout 900 python -m pytest tests/test_gpu_dag.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
