"""profiles/kernel_trace.json: warp_kernel's average duration in the last KEPT rocprofv3 kernel trace, per workload -- what bench.py
prints beside its own HIP-event time (`roofline.trace`).

    python tools/make_trace_json.py r06a        reads profiles/<tag>_kernel_stats_{cfg2,cfg3,cfg4shard}.csv (tools/profile_r05.sh <tag>)
"""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {'cfg2': (300, 1920, 1080), 'cfg3': (600, 1920, 1080), 'cfg4shard': (150, 3840, 2160)}


def main():
    tag = sys.argv[1]
    out = {}
    for wl, (frames, W, H) in SHAPES.items():
        rel = os.path.join('profiles', f'{tag}_kernel_stats_{wl}.csv')
        try:
            rows = list(csv.DictReader(open(os.path.join(REPO, rel))))
        except OSError:
            continue
        for r in rows:
            if 'warp_kernel<true>' in r['Name']:
                out[wl] = {'kernel': 'mf::warp_kernel<true>', 'avg_ms': float(r['AverageNs']) / 1e6, 'min_ms': float(r['MinNs']) / 1e6, 'calls': int(r['Calls']),
                           'algorithmic_bytes_per_launch': 2.0 * frames * W * H * 3, 'frac_of_8TBps': 2.0 * frames * W * H * 3 / (float(r['AverageNs']) * 1e-9) / 8e12,
                           'source': f'{rel} (rocprofv3 --kernel-trace --stats of bench.py --workload {wl}, tools/profile_r05.sh {tag})'}
    with open(os.path.join(REPO, 'profiles', 'kernel_trace.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
    for wl, v in out.items():
        print(f'{wl}: {v["avg_ms"]:.4f} ms avg of {v["calls"]} launches = {v["frac_of_8TBps"]:.4f} of 8 TB/s')


if __name__ == '__main__':
    main()
