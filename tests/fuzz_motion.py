"""Randomised parity campaign for the vertex-motion kernels vs the C oracle (bit-exact or bust).
Usage: python tests/fuzz_motion.py [cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import host, ops
from oracle import clib


def make_case(g):
    W = int(g.integers(8, 4000)); H = int(g.integers(8, 2200))
    R = int(g.integers(1, 40)); C = int(g.integers(1, 40))
    er = int(g.integers(1, 3 * R + 2)); ec = int(g.integers(1, 3 * C + 2))
    P = int(g.integers(1, 6))
    style = g.choice(['uniform', 'outside', 'lattice', 'cluster', 'few'])
    feats, hom = [], np.tile(np.identity(3), (P, 1, 1))
    for p in range(P):
        hom[p, :2, :2] += g.normal(0, 0.01, (2, 2))
        hom[p, :2, 2] = g.normal(0, 5, 2)
        hom[p, 2, :2] = g.normal(0, 1e-6, 2) if g.random() < 0.7 else 0.0
        k = int(g.choice([0, 1, 2, 3, 17, 64, 65, 300, 1200, 2500, 5000], p=[.05, .05, .05, .05, .1, .1, .1, .2, .15, .1, .05]))
        if style == 'few':
            k = min(k, 5)
        if style == 'uniform':
            e = g.random((k, 2)) * [W - 1, H - 1]
        elif style == 'outside':
            e = (g.random((k, 2)) * 1.6 - 0.3) * [W, H]
        elif style == 'lattice':       # exactly on mesh vertices / ellipse extremes: the <= comparisons tie
            e = np.stack([W * g.integers(0, C + 1, k) / C, H * g.integers(0, R + 1, k) / R], -1)
            e = e + g.choice([0.0, 0.0, W / C * ec / 2, -W / C * ec / 2], (k, 1)) * [1, 0]
        else:
            e = g.normal([W / 2, H / 2], [W / 50 + 1, H / 50 + 1], (k, 2))
        e = e.astype(np.float32).astype(np.float64) if g.random() < 0.7 else e
        res = g.normal(0, 2, (k, 2))
        if g.random() < 0.3:
            res = np.round(res)           # many equal residuals: ties in the order statistics
        x, y = e[:, 0], e[:, 1]
        m = hom[p].reshape(9)
        w = x * m[6] + y * m[7] + m[8]
        l = np.stack([(x * m[0] + y * m[1] + m[2]) / w, (x * m[3] + y * m[4] + m[5]) / w], -1) + res
        feats.append((e.reshape(-1, 1, 2), l.reshape(-1, 1, 2)) if k else (None, None))
    return W, H, R, C, er, ec, feats, hom, style


def run(cases=300, seed0=0):
    dev = torch.device('cuda:0')
    g = np.random.default_rng(seed0)
    bad = 0
    t0 = time.time()
    styles = {}
    for case in range(cases):
        W, H, R, C, er, ec, feats, hom, style = make_case(g)
        styles[style] = styles.get(style, 0) + 1
        try:
            want_d, want_v = clib.vertex_motion(W, H, R, C, er, ec, feats, hom, openmp=True)
            want_err = False
        except ValueError:
            want_err = True
        early, late, offsets, kmax = host.pack_features(feats)
        d = [torch.from_numpy(a).to(dev) for a in (early, late, offsets, np.ascontiguousarray(hom))]
        disp, vel, status = ops.vertex_motion(*d, kmax, W, H, R, C, er, ec)
        got_err = bool(status.item())
        ok = got_err == want_err and (want_err or (np.array_equal(disp.cpu().numpy(), want_d) and np.array_equal(vel.cpu().numpy(), want_v)))
        if not ok:
            bad += 1
            print('MISMATCH', case, f'W={W} H={H} R={R} C={C} ellipse={er}x{ec} style={style} K={[0 if e is None else len(e) for e, _ in feats]}', flush=True)
    print(f'{cases} cases in {time.time() - t0:.1f} s ({styles}), mismatches: {bad}')
    return bad


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if run(n, sd) else 0)
