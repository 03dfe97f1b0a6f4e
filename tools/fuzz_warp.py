"""Randomised differential test of the warp path (cell table + plan + warp kernel) against the C oracle, beyond the fixed cases of
tests/: random frame sizes, mesh shapes and motion strengths, including geometries where most footprints take the plan-certified
hot / pair / multi paths (large cells) and ones that stress the general path (small cells, strong jitter).

    python tools/fuzz_warp.py [cases] [seed] [large|tiny]   (needs an MI355X; every case must be bit-identical)
`tiny`: frames of 2-40 x 2-30 pixels, meshes up to the pixel grid and (one case in ten) beyond it -- the end of the size range where the
two bugs of late round 5 sat (no campaign before drew a width below 32).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from meshflow_amd import ops, synthetic
from oracle import clib



def run(cases, seed, verbose=True, large=False, tiny=False):
  rng = np.random.default_rng(seed)
  dev = torch.device('cuda:0')
  bad_total = 0
  for case in range(cases):
      W = int(rng.integers(8, 120)) * 4 if rng.random() < 0.85 else int(rng.integers(30, 400))      # mostly W % 4 == 0 (staged windows)
      H = int(rng.integers(24, 300))
      if large:                                                 # up to 1280 x 720: cells large enough for long runs of certified footprints
          W, H = int(rng.integers(100, 321)) * 4, int(rng.integers(200, 721))
      R, C = int(rng.integers(1, 9)), int(rng.integers(1, 9))
      if rng.random() < 0.2:
          R, C = int(rng.integers(8, 33)), int(rng.integers(8, 33))
      if tiny:
          W, H = int(rng.integers(2, 41)), int(rng.integers(2, 31))
          R, C = int(rng.integers(1, max(2, H))), int(rng.integers(1, max(2, W)))
          if rng.random() < 0.1:
              R, C = int(rng.integers(1, 65)), int(rng.integers(1, 65))       # finer than the pixel grid: degenerate cells, counted
      n = int(rng.integers(1, 4))
      style = rng.random()
      nvert = n * (R + 1) * (C + 1) * 2
      idx = np.arange(nvert, dtype=np.int64).reshape(n, R + 1, C + 1, 2)
      unstab = np.zeros((n, R + 1, C + 1, 2))
      if style < 0.6:          # smooth: a translation per frame + small jitter (what a smoothed path looks like)
          shift = rng.normal(0, rng.uniform(0.5, 6.0), size=(n, 1, 1, 2))
          stab = shift + rng.uniform(0.05, 1.2) * synthetic.normal(idx, seed=1000 + case)
      elif style < 0.9:        # rougher
          stab = rng.uniform(1.0, 4.0) * synthetic.normal(idx, seed=2000 + case)
      else:                    # wild (irregular cells, generic division, overflow lists)
          stab = rng.uniform(5.0, 15.0) * synthetic.normal(idx, seed=3000 + case)
      frames = synthetic.frames_numpy(n, H, W, seed=case, kind='noise')
      want, want_crop, bad = clib.warp_clip(frames, R, C, unstab, stab)
      # three cases in four from / into a stack that does NOT start on a 4-byte boundary (warp_kernel<false>: no staged windows)
      shift = case % 4
      raw_in = torch.zeros(frames.size + 8, dtype=torch.uint8, device=dev)
      raw_out = torch.full((frames.size + 8,), 0xEE, dtype=torch.uint8, device=dev)
      d_fr = raw_in[shift:shift + frames.size].view(frames.shape)
      d_fr.copy_(torch.from_numpy(np.ascontiguousarray(frames)).to(dev))
      try:
          table = ops.cell_table(torch.from_numpy(unstab).to(dev), torch.from_numpy(np.ascontiguousarray(stab)).to(dev), W, H, R, C)
          scanned = ops.crop_scan(table).clone()                  # the scan-only pass (mf_crop_scan_f64) must fill the same values
          out = ops.warp(d_fr, table, (0, 0, 255), out=raw_out[shift:shift + frames.size].view(frames.shape))
          torch.cuda.synchronize()
          table.check()
      except ValueError as e:
          ok = bad != 0
          if verbose or not ok: print(f'case {case}: {W}x{H} mesh {R}x{C} n={n}: degenerate ({e}) -- oracle says {bad}: {"ok" if ok else "MISMATCH"}')
          bad_total += 0 if ok else 1
          continue
      if bad:
          print(f'case {case}: oracle reports {bad} degenerate cells, the HIP path none: MISMATCH')
          bad_total += 1
          continue
      diff = int((out.cpu().numpy() != want).sum())
      cdiff = int((table.crop.cpu().numpy() != want_crop).sum()) + int((scanned.cpu().numpy() != want_crop).sum())
      if diff or cdiff:
          bad_total += 1
          print(f'case {case}: {W}x{H} mesh {R}x{C} n={n} style {style:.2f}: {diff} bytes, {cdiff} crop values differ  <-- MISMATCH')
  return bad_total


if __name__ == '__main__':
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_bad = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0, large='large' in sys.argv[3:], tiny='tiny' in sys.argv[3:])
    print(f'{n_cases} cases, {n_bad} mismatches')
    sys.exit(1 if n_bad else 0)
