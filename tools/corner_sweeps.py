"""Corner sweeps of the rows either side of the path against the NumPy oracle: crop + resize on tiny / odd frames with extreme rectangles,
the stability score on very short clips.      python tools/corner_sweeps.py"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops
from oracle import meshflow_oracle as mo
dev = torch.device('cuda:0')
rng = np.random.default_rng(5)
bad = n = 0
for W, H, nfr in itertools.product((1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 100, 255, 256, 257, 300), (1, 2, 3, 5, 8, 9, 17, 33, 40), (1, 3)):
    frames = rng.integers(0, 256, size=(nfr, H, W, 3), dtype=np.uint8)
    rects = {(0, 0, W - 1, H - 1), (0, 0, 0, 0), (W - 1, H - 1, W - 1, H - 1), (0, H - 1, W - 1, H - 1), (W - 1, 0, W - 1, H - 1)}
    for _ in range(3):
        l, r = sorted(rng.integers(0, W, size=2)); t, b = sorted(rng.integers(0, H, size=2))
        rects.add((int(l), int(t), int(r), int(b)))
    for rect in rects:
        n += 1
        if os.environ.get('VERBOSE'):
            print('case', (W, H, nfr), rect, flush=True)
        want = np.stack(mo.crop_frames(list(frames), rect))
        try:
            got = ops.crop_resize(torch.from_numpy(frames).to(dev), rect).cpu().numpy()
        except Exception as e:
            bad += 1; print('resize ERROR', (W, H, nfr), rect, str(e)[:100]); continue
        if not np.array_equal(got, want):
            bad += 1
            d = np.argwhere(got != want)
            print('resize MISMATCH', (W, H, nfr), rect, len(d), d[:2].tolist())
print(f'crop + resize: {n} cases, {bad} bad')
bad2 = n2 = 0
for F, S in itertools.product((1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 33, 64, 65), (2, 8, 50)):
    stab = np.cumsum(rng.normal(size=(F, S // 2, 1, 2)), axis=0)
    n2 += 1
    try:
        want = mo.stability_score(stab)
    except Exception as e:
        want = ('raises', type(e).__name__)
    try:
        score, _ = ops.stability_score(torch.from_numpy(np.ascontiguousarray(stab)).to(dev))
        got = float(score.item())
    except Exception as e:
        got = ('raises', type(e).__name__)
    ok = (isinstance(want, tuple) and isinstance(got, tuple)) or (not isinstance(want, tuple) and not isinstance(got, tuple) and
                                                                 (abs(got - want) <= 1e-12 * max(1.0, abs(want)) or (np.isnan(got) and np.isnan(want))))
    if not ok:
        bad2 += 1; print('score', (F, S), 'got', got, 'want', want)
print(f'stability score: {n2} cases, {bad2} bad')
sys.exit(1 if bad or bad2 else 0)
