#!/bin/bash
# Round-5 rocprofv3 evidence, one call (tools/profile_r05.sh [tag]): for cfg2 / cfg3 / cfg4shard
#   * the bench line of an unprofiled run with the driver's flags (--gpus 1 --steps 20 --warmup 5)            -> bench_<tag>_<W>.json
#   * rocprofv3 --kernel-trace --stats of bench.py (50 + 5 steps + the spin-up steps, product pipeline only)   -> <tag>_kernel_stats_<W>.csv
#     and the step timeline of that trace (tools/step_timeline.py: per kernel busy time and idle time in front) -> <tag>_step_timeline_<W>.txt
# then the L2-memory-side traffic of warp_kernel (size-resolved read requests + WRITE_SIZE, separate --pmc passes; cfg3 on a 150-frame
# slice: the PMC passes on the 600-frame launch crash rocprofv3 itself) and the SQ counters of warp_kernel at cfg2 (tools/pmc_warp.sh).
# Output: gpurun_out/prof_<tag>/ -- copy what is to be judged into profiles/.
tag=${1:-r05}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O; cd /tmp
for W in cfg2 cfg3 cfg4shard; do
  NF=""; [ $W != cfg2 ] && NF="--no-faithful"
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --workload $W $NF > $O/bench_${tag}_$W.json 2> $O/bench_$W.err
  rm -rf /tmp/ps
  MF_BENCH_NO_EXTRAS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o r -- python3 $R/bench.py --steps 50 --warmup 5 --cpu-frames 0 --no-e2e --no-workloads --workload $W > $O/stats_$W.log 2>&1
  grep -E "\"Name\"|mf::" /tmp/ps/r_kernel_stats.csv > $O/${tag}_kernel_stats_$W.csv
  python3 $R/tools/step_timeline.py /tmp/ps/r_kernel_trace.csv 30 > $O/${tag}_step_timeline_$W.txt 2>&1
done
: > $O/${tag}_traffic_rdreq.csv
for W in cfg2 cfg4shard cfg3; do
  FR=""; [ $W = cfg3 ] && FR="--frames 150"
  for set in "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum" "WRITE_SIZE"; do
    rm -rf /tmp/tr
    MF_BENCH_NO_EXTRAS=1 timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/tr -o r -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-e2e --no-workloads --workload $W $FR > /tmp/tr.log 2>&1
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
cd $R && bash tools/pmc_warp.sh meshflow_amd/libmeshflow_hip.so cfg2 $tag > /dev/null 2>&1; cp gpurun_out/pmc_$tag/summary.csv $O/${tag}_sq_warp.csv
for W in cfg2 cfg3 cfg4shard; do head -4 $O/${tag}_kernel_stats_$W.csv | cut -c1-60,200-330; cat $O/${tag}_step_timeline_$W.txt; done
