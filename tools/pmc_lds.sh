#!/bin/bash
# LDS / wait counters of one kernel (KERNEL=...) under bench.py
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES"; do
  rm -rf /tmp/pl
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pl -o r -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-frames 0 > /tmp/pl.log 2>&1
  grep "${KERNEL:-warp_kernel}" /tmp/pl/r_counter_collection.csv | tail -8 | awk -F',' '{n=NF; print $(n-3), $(n-2), $(n)-$(n-1)}'
done
