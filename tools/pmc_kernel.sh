#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <kernel-name substring> <python script + args ...>
# SQ counters of every kernel whose name contains the substring, per launch (mean), in separate rocprofv3 --pmc passes of <= 8 counters
# (--kernel-trace only next to --pmc: the pool refuses anything else).  Output: gpurun_out/pmc_<tag>/summary.csv
tag=$1; pat=$2; shift 2
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$tag; mkdir -p $O; cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  i=$((i+1)); rm -rf /tmp/pk$i
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pk$i -o r -- python3 "$@" > $O/run_$i.log 2>&1
  python3 - /tmp/pk$i/r_counter_collection.csv "$pat" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[2] in r['Kernel_Name']:
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            rows[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
except FileNotFoundError:
    print('no output for this pass')
for (k, c), v in sorted(rows.items()):
    print(f'{k},{c},{len(v)},{sum(v) / len(v):.0f}')
PY
done | tee $O/summary.csv
