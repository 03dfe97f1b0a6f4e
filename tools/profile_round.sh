#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box and leaves small summaries in gpurun_out/prof_<tag>/.
#   1. kernel-trace + stats of `bench.py` (cfg2)
#   2. PMC passes FETCH_SIZE and WRITE_SIZE (separately: TCC has 4 slots) of the same command
#   3. the same two PMC passes on tools/calib_fetch (known byte counts) to calibrate the counters
tag=${1:-r01}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O; cd /tmp
WORKLOAD=${WORKLOAD:-cfg2}
BENCH="python3 $R/bench.py --steps 5 --warmup 1 --cpu-frames 0 --no-e2e --workload $WORKLOAD"
# >= 50 launches per kernel for the duration statistics (the PMC passes serialise kernels and need fewer)
BENCH_STATS="python3 $R/bench.py --steps ${STATS_STEPS:-50} --warmup 5 --cpu-frames 0 --no-e2e --workload $WORKLOAD"
rm -rf /tmp/ps /tmp/pf /tmp/pw /tmp/cf /tmp/cw
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o r -- $BENCH_STATS > $O/stats_run.log 2>&1
grep -E "\"Name\"|mf::" /tmp/ps/r_kernel_stats.csv > $O/kernel_stats.csv
cp /tmp/ps/r_domain_stats.csv $O/ 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o r -- $BENCH > $O/fetch_run.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o r -- $BENCH > $O/write_run.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cf -o r -- $R/tools/calib_fetch > $O/calib_fetch_run.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/cw -o r -- $R/tools/calib_fetch > $O/calib_write_run.log 2>&1
python3 - "$O" <<'PY'
import csv, sys, collections
O = sys.argv[1]
def load(path, pat):
    rows = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open(path)):
            name = r['Kernel_Name']
            if any(p in name for p in pat):
                short = name.split('(')[0].replace('void ', '')
                rows[(short, r['Counter_Name'])].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    except FileNotFoundError:
        pass
    return rows
with open(O + '/pmc_summary.csv', 'w') as out:
    out.write('source,kernel,counter,launches,mean_value_KiB,mean_duration_ns\n')
    for src, path in (('bench', '/tmp/pf/r_counter_collection.csv'), ('bench', '/tmp/pw/r_counter_collection.csv'),
                      ('calib', '/tmp/cf/r_counter_collection.csv'), ('calib', '/tmp/cw/r_counter_collection.csv')):
        for (k, c), v in sorted(load(path, ['mf::', 'calib_']).items()):
            out.write('%s,%s,%s,%d,%.1f,%.0f\n' % (src, k, c, len(v), sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v)))
print(open(O + '/pmc_summary.csv').read())
PY
cat $O/kernel_stats.csv | cut -c1-200
