"""Randomised parity campaign: HIP kernels vs the C oracle on random geometries (bit-exact or bust).
Usage: python tests/fuzz_parity.py [cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic
from oracle import clib, meshflow_oracle as mo



def run(cases=200, seed0=0, only=-1, big=False):
    """Returns (number of mismatching cases, per-kind counts)."""
    dev = torch.device('cuda:0')
    g = np.random.default_rng(seed0)
    bad_cases = 0
    t_start = time.time()
    stats = {'warp': 0, 'warp_degenerate': 0, 'jacobi': 0, 'resize': 0}
    for case in range(cases):
        kind = g.choice(['warp', 'warp', 'warp', 'jacobi', 'resize'])
        if kind == 'warp':
            W = int(g.integers(8, 2100 if big else 420)); H = int(g.integers(8, 1200 if big else 300))
            if g.random() < 0.6:
                W = max(8, W & ~3)                               # the LDS-staged tap path needs W % 4 == 0
            R = int(g.integers(1, min(64 if big else 24, H // 2) + 1)); C = int(g.integers(1, min(64 if big else 24, W // 2) + 1))
            n = int(g.integers(1, 3 if big else 4))
            sigma = float(g.choice([0.2, 1.0, 3.0, 8.0, 25.0])) * min(1.0, min(W / C, H / R) / 20.0 + 0.05)
            frames = synthetic.frames_numpy(n, H, W, seed=case, kind='noise')
            idx = np.arange(n * (R + 1) * (C + 1) * 2, dtype=np.int64).reshape(n, R + 1, C + 1, 2)
            unstab = 2.0 * synthetic.normal(idx, seed=2 * case + 1)
            stab = unstab + sigma * synthetic.normal(idx, seed=2 * case + 2)
            if g.random() < 0.3:                                  # add a global shift so that borders are exercised
                stab = stab + g.normal(0, 6, (n, 1, 1, 2))
            border = tuple(int(v) for v in g.integers(0, 256, 3))
            use_bbox = bool(g.random() < 0.5)
            if only >= 0 and case != only:
                continue
            want, want_crop, bad = clib.warp_clip(frames, R, C, unstab, stab, border_bgr=border, use_bbox=use_bbox or big, openmp=big)
            table = ops.cell_table(torch.from_numpy(unstab).to(dev), torch.from_numpy(stab).to(dev), W, H, R, C)
            out = ops.warp(torch.from_numpy(frames).to(dev), table, border)
            torch.cuda.synchronize()
            nbad = int(table.status.item())
            ok = nbad == bad
            if bad == 0:
                ok = ok and np.array_equal(out.cpu().numpy(), want) and np.array_equal(table.crop.cpu().numpy(), want_crop)
                stats['warp'] += 1
            else:
                stats['warp_degenerate'] += 1
            desc = f'warp W={W} H={H} R={R} C={C} n={n} sigma={sigma:.2f} degenerate={bad}'
            if only >= 0:
                o = out.cpu().numpy()
                for f in range(n):
                    d = np.argwhere((o[f] != want[f]).any(axis=2))
                    print('frame', f, 'differing pixels', len(d), d[:20].tolist())
                    for (yy, xx) in d[:5]:
                        print('   at', yy, xx, 'got', o[f][yy, xx], 'want', want[f][yy, xx])
                print('crop got', table.crop.cpu().numpy().tolist(), 'want', want_crop.tolist())
                tab, _ = clib.cell_table(W, H, R, C, unstab[0], stab[0])
                print('records equal (frame 0):', np.array_equal(table.records().cpu().numpy()[0], tab))
        elif kind == 'jacobi':
            if only >= 0: 
                # keep the random stream in step
                pass
            F = int(g.integers(1, 900)); S = int(g.integers(1, 40)); omega = int(g.choice([1, 2, 5, 10, 10, 30, 30, 17, 40])); iters = int(g.integers(0, 40))
            b = np.cumsum(3.0 * synthetic.normal(np.arange(F * S).reshape(F, S), seed=case), axis=0)
            taps = np.exp(-np.square((3 / omega) * np.arange(-omega, omega + 1)))
            lam = 0.95 * synthetic.uniform01(np.arange(F), seed=case + 7)
            inv_on = 1.0 / (1 + 2 * lam * taps.sum())
            want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            got = ops.jacobi(t(b), t(taps), t(lam), t(inv_on), omega, iters).cpu().numpy()
            ok = np.array_equal(got, want)
            stats['jacobi'] += 1
            desc = f'jacobi F={F} S={S} omega={omega} iters={iters}'
        else:
            W = int(g.integers(4, 300)); H = int(g.integers(4, 200)); n = int(g.integers(1, 3))
            l = int(g.integers(0, W)); r = int(g.integers(l, W)); tp = int(g.integers(0, H)); bt = int(g.integers(tp, H))
            frames = synthetic.frames_numpy(n, H, W, seed=case, kind='noise')
            want = np.stack(mo.crop_frames(list(frames), (l, tp, r, bt)))
            got = ops.crop_resize(torch.from_numpy(frames).to(dev), (l, tp, r, bt)).cpu().numpy()
            ok = np.array_equal(got, want)
            stats['resize'] += 1
            desc = f'resize W={W} H={H} crop=({l},{tp},{r},{bt})'
        if not ok:
            bad_cases += 1
            print('MISMATCH', case, desc, flush=True)
    print(f'{cases} cases in {time.time() - t_start:.1f} s: {stats}, mismatches: {bad_cases}')
    return bad_cases, stats


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    one = int(sys.argv[3]) if len(sys.argv) > 3 else -1            # re-run just this case, verbosely
    big = len(sys.argv) > 4 and sys.argv[4] == 'big'                # frames up to 2100 x 1200, meshes up to 64 x 64
    sys.exit(1 if run(n, sd, one, big)[0] else 0)
