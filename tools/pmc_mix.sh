#!/bin/bash
# usage: tools/pmc_mix.sh <tag> -- runs two PMC passes of bench.py and prints per-launch counters of warp_kernel
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/prof; cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pp$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pp$i -o r -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-frames 0 > $O/prof/pmc_$1_$i.log 2>&1
  grep "${KERNEL:-warp_kernel}" /tmp/pp$i/r_counter_collection.csv | tail -8 | awk -F',' '{n=NF; print $(n-3), $(n-2), $(n)-$(n-1)}'
done
