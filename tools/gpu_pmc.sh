#!/bin/bash
# PMC passes for the sweep kernel (variant in $1, tag in $2)
V=${1:-0}; TAG=${2:-pmc}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B="python $PWD/bench.py --no-cpu-baseline --m-per-gpu 131072 --steps 1 --warmup 0 --variant $V"
( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --kernel-trace -d $OUT/prof_${TAG}_a -o a -- $B > $OUT/pmc_a.log 2>&1 ); echo "a rc=$?"
( cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/prof_${TAG}_b -o b -- $B > $OUT/pmc_b.log 2>&1 ); echo "b rc=$?"
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/prof_${TAG}_c -o c -- $B > $OUT/pmc_c.log 2>&1 ); echo "c rc=$?"
