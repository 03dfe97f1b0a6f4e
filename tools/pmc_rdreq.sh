#!/bin/bash
# Read-request size mix at the L2's memory side (TCC_EA0_RDREQ by size) for the calibration kernels and bench.py.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for what in calib bench; do
  if [ $what = calib ]; then CMD="$R/tools/calib_fetch"; else CMD="python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0"; fi
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
    rm -rf /tmp/rq
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/rq -o r -- $CMD > /tmp/rq.log 2>&1
    python3 - <<'PY'
import csv, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/rq/r_counter_collection.csv')):
    n = r['Kernel_Name']
    if 'calib_' in n or 'warp_kernel' in n or 'resize_kernel' in n:
        rows[(n.split('(')[0].replace('void ', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(rows.items()):
    print(f'{k},{c},{len(v)},{sum(v) / len(v):.0f}')
PY
  done
done
