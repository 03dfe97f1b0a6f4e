#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/pq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT --output-format csv -d /tmp/pq -o r -- python3 $R/tools/single_owner_probe.py > /tmp/pq.log 2>&1
tail -1 /tmp/pq.log
grep "warp_kernel" /tmp/pq/r_counter_collection.csv | tail -8 | awk -F',' '{n=NF; printf "%s %.1f\n", $(n-3), $(n-2)/9792000}'
