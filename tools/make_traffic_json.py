"""profiles/traffic.json from the size-resolved read-request / WRITE_SIZE passes of tools/profile_r04.sh.

    python tools/make_traffic_json.py profiles/r03e_traffic_rdreq.csv "end of round 3"
"""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {'cfg2': (300, 1920, 1080, '16x16', 1), 'cfg4shard': (150, 3840, 2160, '16x16', 1), 'cfg3': (150, 1920, 1080, '32x32', 4)}
METHOD = ('reads = 128*TCC_EA0_RDREQ_128B + 64*TCC_EA0_RDREQ_64B + 32*TCC_EA0_RDREQ_32B (sum over channels): checked on tools/calib_fetch.hip '
          'calib_wide16, where it returns the 1,866,240,000 bytes read exactly; the derived FETCH_SIZE tallies every request at 64 B on gfx950 '
          'and reads half of that (the guide\'s x2 rule). writes = WRITE_SIZE (KiB, exact on the same calibration kernel). Separate --pmc '
          'passes (tools/profile_r05.sh), per launch of mf::warp_kernel, at the HEAD kernels.')


def main():
    path, when = sys.argv[1], sys.argv[2]
    vals = {}
    for w, counter, launches, mean in csv.reader(open(path)):
        vals.setdefault(w, {})[counter] = float(mean)
    out = {}
    for w, (frames, W, H, mesh, scale) in SHAPES.items():
        v = vals[w]
        reads = 128 * v['TCC_EA0_RDREQ_128B_sum'] + 64 * v['TCC_EA0_RDREQ_64B_sum'] + 32 * v['TCC_EA0_RDREQ_32B_sum']
        writes = 1024 * v['WRITE_SIZE']
        algo = 2 * frames * W * H * 3 * scale
        src = f'{os.path.relpath(path, REPO)} ({when}), {frames} frames of {W}x{H}, {mesh} mesh'
        if scale != 1:
            src = (f'{os.path.relpath(path, REPO)} ({when}), a {frames}-frame SLICE of the {w} clip (bench.py --workload {w} --frames {frames}: same '
                   f'geometry, mesh and smoothed motion), scaled x{scale} to the {frames * scale}-frame launch')
        out[w] = {'source': src, 'read_bytes': round(reads * scale), 'write_bytes': round(writes * scale),
                  'traffic_bytes': round((reads + writes) * scale), 'algorithmic_bytes': algo,
                  'ratio': round((reads + writes) * scale / algo, 4), 'method': METHOD}
        if scale != 1:
            out[w]['note'] = ('the PMC passes on the full 600-frame launch crash rocprofv3 itself (segmentation fault in the profiler process, '
                              'rounds 1-2); per-frame traffic does not depend on the number of frames in a launch')
    with open(os.path.join(REPO, 'profiles', 'traffic.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
    for w in out:
        print(w, out[w]['traffic_bytes'], out[w]['ratio'])


if __name__ == '__main__':
    main()
