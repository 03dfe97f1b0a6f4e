#!/bin/bash
# usage: tools/pmc_warp.sh <lib.so> <workload> <tag>  -- SQ / LDS counters of warp_kernel for one build (separate passes of 8 counters)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$3; mkdir -p $O; cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_THREAD_CYCLES_VALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  i=$((i+1)); if [ -n "$SETS" ] && [ $i -gt $SETS ]; then break; fi; rm -rf /tmp/pw$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pw$i -o r -- python3 $R/tools/ab_warp.py --workloads $2 --rounds 1 --launches 2 $R/$1 > $O/run_$i.log 2>&1
  python3 - /tmp/pw$i/r_counter_collection.csv <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if 'warp_kernel' in r['Kernel_Name']:
            rows[r['Counter_Name']].append(float(r['Counter_Value']))
except FileNotFoundError:
    print('no output for this pass')
for k, v in rows.items():
    print(f'{k},{len(v)},{sum(v) / len(v):.0f}')
PY
done | tee $O/summary.csv
