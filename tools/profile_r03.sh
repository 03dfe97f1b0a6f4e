#!/bin/bash
# Round-3 rocprofv3 evidence, one call: for cfg2 / cfg3 / cfg4shard the kernel-trace stats of bench.py (50 + 5 steps, + the spin-up steps) and the bench
# line of an unprofiled run with the driver's flags (--gpus 1 --steps 20 --warmup 5); then the L2-memory-side traffic of warp_kernel (size-resolved read requests + WRITE_SIZE, separate --pmc
# passes; cfg3 on a 150-frame slice: the PMC passes on the 600-frame launch crash rocprofv3 itself).  Output: gpurun_out/prof_r03/.
tag=${1:-r03}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O; cd /tmp
for W in cfg2 cfg3 cfg4shard; do
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --workload $W --no-faithful > $O/bench_${tag}_$W.json 2> $O/bench_$W.err
  rm -rf /tmp/ps
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o r -- python3 $R/bench.py --steps 50 --warmup 5 --cpu-frames 0 --no-e2e --workload $W > $O/stats_$W.log 2>&1
  grep -E "\"Name\"|mf::" /tmp/ps/r_kernel_stats.csv > $O/${tag}_kernel_stats_$W.csv
done
: > $O/${tag}_traffic_rdreq.csv
for W in cfg2 cfg4shard cfg3; do
  FR=""; [ $W = cfg3 ] && FR="--frames 150"
  for set in "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum" "WRITE_SIZE"; do
    rm -rf /tmp/tr
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/tr -o r -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-e2e --workload $W $FR > /tmp/tr.log 2>&1
    python3 - "$W" >> $O/${tag}_traffic_rdreq.csv <<'PY'
import csv, collections, sys
rows = collections.defaultdict(list)
try:
    for r in csv.DictReader(open('/tmp/tr/r_counter_collection.csv')):
        if 'warp_kernel' in r['Kernel_Name']:
            rows[r['Counter_Name']].append(float(r['Counter_Value']))
except FileNotFoundError:
    print(f'{sys.argv[1]},FAILED,0,0')
for c, v in sorted(rows.items()):
    print(f'{sys.argv[1]},{c},{len(v)},{sum(v) / len(v):.1f}')
PY
  done
done
cat $O/${tag}_traffic_rdreq.csv
for W in cfg2 cfg3 cfg4shard; do head -3 $O/${tag}_kernel_stats_$W.csv | cut -c1-60,200-330; done
