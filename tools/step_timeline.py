"""Timeline of the kernels of bench.py's timed steps from a rocprofv3 --kernel-trace CSV: per kernel the mean duration and the mean idle
gap in front of it (end of the previous kernel on any stream -> its start), over the steps between the first and last warp launch.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-e2e
    python tools/step_timeline.py /tmp/kt/r_kernel_trace.csv [skip_first_n_warps]
"""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')) for r in rows))
warps = [i for i, k in enumerate(ks) if 'warp_kernel' in k[2]]
lo, hi = warps[skip], warps[-1]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
busy_until = ks[lo][1]
for s, e, name in ks[lo + 1:hi + 1]:
    short = name[:60]
    dur[short] += (e - s) / 1e3
    gap[short] += max(0, s - busy_until) / 1e3
    cnt[short] += 1
    busy_until = max(busy_until, e)
steps = len([i for i in warps if lo < i <= hi])
print(f'{steps} steps, {(ks[hi][1] - ks[lo][1]) / 1e3 / steps:.1f} us per step (end of warp to end of warp)')
for k in dur:
    print(f'{k:62s} x{cnt[k] / steps:4.1f}/step   {dur[k] / steps:8.1f} us busy   {gap[k] / steps:6.1f} us idle in front')
print(f'{"sum":62s}            {sum(dur.values()) / steps:8.1f}            {sum(gap.values()) / steps:6.1f}')
