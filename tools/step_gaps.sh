#!/bin/bash
# Kernel timeline of bench.py steps: durations and the idle gaps between consecutive kernels (rocprofv3 kernel trace).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python3 $R/bench.py --steps 8 --warmup 2 --cpu-frames 0 --no-e2e > /tmp/kt.log 2>&1
python3 - <<'PY'
import csv
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mf::', '')[:28])
               for r in csv.DictReader(open('/tmp/kt/r_kernel_trace.csv'))), key=lambda t: t[0])
# find the timed steps: sequences jacobi -> cell_table -> plan -> warp -> crop_reduce
idx = [i for i, r in enumerate(rows) if r[2].startswith('jacobi_wave_kernel')]
for start in idx[4:8]:
    prev_end = rows[start - 1][1]
    line = []
    for s, e, n in rows[start:start + 5]:
        line.append(f'gap {(s - prev_end) / 1e3:5.1f} us | {n} {(e - s) / 1e3:7.1f} us')
        prev_end = e
    print('\n'.join(line)); print('---')
PY
