#!/bin/bash
# L2-memory-side traffic of warp_kernel for one bench workload (WORKLOAD=cfg2|cfg3|cfg4shard): size-resolved read requests + WRITE_SIZE.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; W=${WORKLOAD:-cfg2}
for set in "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum" "WRITE_SIZE"; do
  rm -rf /tmp/tr
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/tr -o r -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-e2e --workload $W ${FRAMES:+--frames $FRAMES} > /tmp/tr.log 2>&1
  python3 - "$W" <<'PY'
import csv, collections, sys
rows = collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/tr/r_counter_collection.csv')):
    if 'warp_kernel' in r['Kernel_Name']:
        rows[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(rows.items()):
    print(f'{sys.argv[1]},{c},{len(v)},{sum(v) / len(v):.1f}')
PY
done
