"""-m gpu: a THIRD-PARTY second opinion on the restated OpenCV calls, with what the GPU box has: torch.

No OpenCV exists in the build image or on the GPU box, so the five cv2 calls on the path (findHomography, warpPerspective,
perspectiveTransform, remap: mfs.py:1041-1069; resize: mfs.py:1150) are restated from OpenCV's published algorithms and the
NumPy oracle, the C oracle and the HIP kernels all share that reading ("parity unpinned").  These tests
compare the HIP path with implementations that share NOTHING with it:

  * the warp with `torch.nn.functional.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True)` fed a coordinate
    map computed here in float64 -- pixel centres at integers, the map read as "output pixel -> source position", taps outside
    the frame replaced by the (black) border colour;
  * `_crop_frames` with `torch.nn.functional.interpolate(mode='bilinear', align_corners=False, antialias=False)` on the crop --
    half-pixel centres.

cv2.remap quantises coordinates to 1/32 pixel and cv2.resize its weights to 11 bits, so the comparison allows 1 LSB (north_star's
own bar) around the envelope of the float64 result over a 1/32-pixel neighbourhood; identity and integer shifts must be exact.
It pins nothing (parity stays "partial": tests/test_cv2_crosscheck.py is the door to "green"), but a wrong reading of pixel
centres, map direction, mesh-motion sign or border handling cannot pass it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
F_ = torch.nn.functional


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return torch.device('cuda:0')


def _grid(W, H, R, C):
    """Mesh vertex pixel positions (mfs.py:881-906): ceil((W-1) col / C), ceil((H-1) row / R)."""
    gx = np.array([np.ceil((W - 1) * (c / C)) for c in range(C + 1)])
    gy = np.array([np.ceil((H - 1) * (r / R)) for r in range(R + 1)])
    return gx, gy


def _smooth_frames(dev, n, H, W, seed=0):
    """Band-limited frames (sums of sinusoids, gradients of a few grey levels per pixel): uint8 (n, H, W, 3) on the device."""
    y = torch.arange(H, dtype=torch.float64, device=dev)[None, :, None, None]
    x = torch.arange(W, dtype=torch.float64, device=dev)[None, None, :, None]
    c = torch.arange(3, dtype=torch.float64, device=dev)[None, None, None, :]
    f = torch.arange(n, dtype=torch.float64, device=dev)[:, None, None, None] + seed
    v = 128 + 60 * torch.sin(0.031 * x + 0.017 * y + 0.7 * c + 0.3 * f) + 50 * torch.cos(0.011 * x - 0.043 * y + 1.3 * c - 0.2 * f)
    return v.round().clamp(0, 255).to(torch.uint8)


def _sample(frames, u, v):
    """grid_sample of uint8 frames (n, H, W, 3) at float64 source positions u, v (n, H, W) in PIXEL units -> float64 (n, H, W, 3)."""
    n, H, W, _ = frames.shape
    src = frames.permute(0, 3, 1, 2).to(torch.float64)
    grid = torch.stack([2 * u / (W - 1) - 1, 2 * v / (H - 1) - 1], dim=-1)          # align_corners=True: -1 <-> pixel 0, +1 <-> pixel W-1
    return F_.grid_sample(src, grid, mode='bilinear', padding_mode='zeros', align_corners=True).permute(0, 2, 3, 1)


def _assert_within_envelope(got, frames, u, v, allow=1.0, skip=None):
    """got (uint8) must lie within `allow` grey levels of the range the float64 bilinear sample takes over the 1/32-pixel bucket
    around (u, v) (cv2.remap rounds 32 u, 32 v to integers)."""
    lo = hi = None
    for du in (-1 / 64, 0.0, 1 / 64):
        for dv in (-1 / 64, 0.0, 1 / 64):
            s = _sample(frames, u + du, v + dv)
            lo = s if lo is None else torch.minimum(lo, s)
            hi = s if hi is None else torch.maximum(hi, s)
    g = got.to(torch.float64)
    bad = (g < lo - allow - 0.5) | (g > hi + allow + 0.5)                           # (+0.5: rounding of the final value)
    if skip is not None:
        bad = bad & ~skip[..., None]
    assert not bool(bad.any()), f'{int(bad.sum())} of {bad.numel()} values outside the envelope; worst {float(torch.maximum(lo - g, g - hi).max()):.2f}'
    return float((g - _sample(frames, u, v)).abs().mean())


def _hip_warp(dev, d_frames, R, C, unstab, stab, border=(0, 0, 0)):
    from meshflow_amd import ops
    n, H, W, _ = d_frames.shape
    table = ops.cell_table(torch.from_numpy(np.ascontiguousarray(unstab)).to(dev), torch.from_numpy(np.ascontiguousarray(stab)).to(dev), W, H, R, C)
    out = ops.warp(d_frames, table, border)
    torch.cuda.synchronize()
    table.check()
    return out, table.crop.cpu().numpy()


def _pixels(dev, n, H, W):
    ys = torch.arange(H, dtype=torch.float64, device=dev)[None, :, None].expand(n, H, W)
    xs = torch.arange(W, dtype=torch.float64, device=dev)[None, None, :].expand(n, H, W)
    return xs, ys


@pytest.mark.parametrize('H,W,R,C', [(360, 640, 16, 16), (270, 484, 5, 7), (1080, 1920, 16, 16)])
def test_identity_and_integer_shifts_are_exact(dev, H, W, R, C):
    """No motion: the frame itself.  Stabilized = unstabilized + (dx, dy): the content moves by exactly (+dx, +dy) (content at grid
    vertex v moves to v + (P - C), mfs.py:964-967), the uncovered band is the border colour -- grid_sample with the map
    (x - dx, y - dy) says the same, byte for byte."""
    n = 2
    frames = _smooth_frames(dev, n, H, W)
    z = np.zeros((n, R + 1, C + 1, 2))
    out, crop = _hip_warp(dev, frames, R, C, z, z)
    assert torch.equal(out, frames)
    xs, ys = _pixels(dev, n, H, W)
    for dx, dy in ((5, -3), (-4, 6)):
        s = z.copy(); s[..., 0] = dx; s[..., 1] = dy
        out, crop = _hip_warp(dev, frames, R, C, z, s)
        want = _sample(frames, xs - dx, ys - dy)                         # integer positions: weights 0 / 1
        # one ring of pixels next to the uncovered band: cv2.warpPerspective's bilinear mask reaches one pixel beyond the mesh
        # (those pixels sample column -1 / W: black either way) -- equal as well
        # (grid_sample goes through normalised coordinates: an integer position comes back as integer +- 1e-13, the weights as
        # 1 - 1e-13 and 1e-13 -- "exact" is |difference| < 1e-6 grey levels on every byte)
        assert float((out.to(torch.float64) - want).abs().max()) < 1e-6
        assert crop.tolist() == [[max(dx, 0), max(dy, 0), W - 1 + min(dx, 0), H - 1 + min(dy, 0)]] * n


@pytest.mark.parametrize('H,W,R,C,seed', [(360, 640, 16, 16, 1), (270, 484, 5, 7, 2), (1080, 1920, 16, 16, 3), (1080, 1920, 32, 32, 4)])
def test_global_homography_vs_grid_sample(dev, H, W, R, C, seed):
    """Every vertex moved by ONE global homography G (rotation, shear, perspective, sub-pixel shift): every cell's 4-point
    homography is G, whichever cell owns a pixel, so the whole frame is out(x) = src(G^-1 x) -- computed here in float64 with
    torch.linalg, sampled by grid_sample.  Pixel centres, map direction, the sign of the mesh motion and the constant border all
    have to agree; the only licence is cv2.remap's 1/32-pixel coordinate bucket + 1 LSB."""
    rng = np.random.RandomState(seed)
    n = 2
    frames = _smooth_frames(dev, n, H, W, seed)
    gx, gy = _grid(W, H, R, C)
    unstab = np.zeros((n, R + 1, C + 1, 2))
    stab = np.zeros_like(unstab)
    Gs = []
    for f in range(n):
        a = rng.uniform(-0.01, 0.01)
        G = np.array([[np.cos(a) * (1 + rng.uniform(-0.01, 0.01)), -np.sin(a) + rng.uniform(-0.004, 0.004), rng.uniform(-6, 6)],
                      [np.sin(a), np.cos(a) * (1 + rng.uniform(-0.01, 0.01)), rng.uniform(-6, 6)],
                      [rng.uniform(-4e-6, 4e-6), rng.uniform(-4e-6, 4e-6), 1.0]])
        Gs.append(G)
        X, Y = np.meshgrid(gx, gy)                                       # (R+1, C+1)
        w = G[2, 0] * X + G[2, 1] * Y + G[2, 2]
        stab[f, :, :, 0] = (G[0, 0] * X + G[0, 1] * Y + G[0, 2]) / w - X
        stab[f, :, :, 1] = (G[1, 0] * X + G[1, 1] * Y + G[1, 2]) / w - Y
    out, _ = _hip_warp(dev, frames, R, C, unstab, stab)
    xs, ys = _pixels(dev, n, H, W)
    Gi = torch.linalg.inv(torch.from_numpy(np.stack(Gs)).to(dev))      # (n, 3, 3) float64
    g = lambda i, j: Gi[:, i, j][:, None, None]
    w = g(2, 0) * xs + g(2, 1) * ys + g(2, 2)
    u = (g(0, 0) * xs + g(0, 1) * ys + g(0, 2)) / w
    v = (g(1, 0) * xs + g(1, 1) * ys + g(1, 2)) / w
    # Outside the warped mesh the reference paints the border colour where grid_sample still blends the frame's edge pixels with
    # black over one pixel: skip the ring of output pixels whose source lies within 1.5 pixels outside the frame
    ring = ((u < 0) & (u > -1.5)) | ((u > W - 1) & (u < W + 0.5)) | ((v < 0) & (v > -1.5)) | ((v > H - 1) & (v < H + 0.5))
    # (the vertices are float32 when cv2.findHomography sees them, mfs.py:1041: 1e-4 px at these coordinates -- inside the envelope)
    mean_abs = _assert_within_envelope(out, frames, u, v, allow=1.0, skip=ring)
    assert mean_abs < 0.6                                                # typical distance from the float64 value: rounding + bucket
    far = (u < -2) | (u > W + 1) | (v < -2) | (v > H + 1)                # well outside the frame: the border colour (black)
    assert bool((out[far] == 0).all())


def test_mesh_motion_vs_grid_sample_on_cell_interiors(dev):
    """Real mesh motion (every cell its own homography): per cell the exact 4-point homography is solved here with
    torch.linalg.solve (stabilized corners -> grid corners), and every output pixel that lies INSIDE the stabilized quad of
    exactly one cell, two pixels away from its edges, must be that cell's map sampled by grid_sample.  (Pixels near cell borders
    are owned by the painter order of mfs.py:1031-1061 -- pinned by the goldens, not judged here.)"""
    from meshflow_amd import synthetic
    H, W, R, C, n = 360, 640, 8, 8, 2
    frames = _smooth_frames(dev, n, H, W, 5)
    disp, hom = synthetic.motion(n + 6, R, C, seed=7, translation_sigma=2.0, field_sigma=1.5)
    unstab = disp[3:3 + n]
    stab = unstab + 0.6 * (disp[5:5 + n] - unstab)                       # some smooth per-vertex motion
    out, _ = _hip_warp(dev, frames, R, C, unstab, stab)
    gx, gy = _grid(W, H, R, C)
    xs, ys = _pixels(dev, 1, H, W)
    xs, ys = xs[0], ys[0]
    for f in range(n):
        u = torch.full((H, W), float('nan'), dtype=torch.float64, device=dev)
        v = torch.full_like(u, float('nan'))
        owners = torch.zeros((H, W), dtype=torch.int32, device=dev)
        P = np.stack(np.meshgrid(gx, gy), axis=-1) + (stab[f] - unstab[f])          # stabilized vertex positions (R+1, C+1, 2)
        for r in range(R):
            for c in range(C):
                src = np.array([P[r, c], P[r, c + 1], P[r + 1, c], P[r + 1, c + 1]]).astype(np.float32).astype(np.float64)
                dst = np.array([[gx[c], gy[r]], [gx[c + 1], gy[r]], [gx[c], gy[r + 1]], [gx[c + 1], gy[r + 1]]])
                A, b = [], []
                for (x, y), (X, Y) in zip(src, dst):                     # h maps (x, y) -> (X, Y), h22 = 1
                    A.append([x, y, 1, 0, 0, 0, -X * x, -X * y]); b.append(X)
                    A.append([0, 0, 0, x, y, 1, -Y * x, -Y * y]); b.append(Y)
                h = torch.linalg.solve(torch.tensor(A, dtype=torch.float64), torch.tensor(b, dtype=torch.float64))
                h = [float(t) for t in h] + [1.0]
                w = h[6] * xs + h[7] * ys + h[8]
                uu = (h[0] * xs + h[1] * ys + h[2]) / w
                vv = (h[3] * xs + h[4] * ys + h[5]) / w
                inside = (uu > gx[c] + 2) & (uu < gx[c + 1] - 2) & (vv > gy[r] + 2) & (vv < gy[r + 1] - 2)
                loose = (uu > gx[c] - 2) & (uu < gx[c + 1] + 2) & (vv > gy[r] - 2) & (vv < gy[r + 1] + 2)
                owners += loose.to(torch.int32)
                u = torch.where(inside, uu, u)
                v = torch.where(inside, vv, v)
        sure = ~torch.isnan(u) & (owners == 1)
        assert float(sure.double().mean()) > 0.6                          # most of the frame is judged
        u = torch.where(sure, u, torch.zeros_like(u)); v = torch.where(sure, v, torch.zeros_like(v))
        _assert_within_envelope(out[f:f + 1], frames[f:f + 1], u[None], v[None], allow=1.0, skip=~sure[None])


@pytest.mark.parametrize('H,W,rect', [(360, 640, (13, 11, 629, 350)), (1080, 1920, (17, 9, 1899, 1071)), (270, 484, (0, 0, 483, 269)),
                                      (270, 484, (40, 30, 443, 239))])
def test_crop_resize_vs_interpolate(dev, H, W, rect):
    """_crop_frames (mfs.py:1111-1157: crop to the inclusive rectangle, cv2.resize back to (W, H), INTER_LINEAR) against
    torch's bilinear interpolate with half-pixel centres (align_corners=False), float64: within 1 LSB (cv2 quantises the
    weights to 11 bits and truncates twice); the full-frame rectangle is the identity."""
    from meshflow_amd import ops
    frames = _smooth_frames(dev, 2, H, W, 9)
    left, top, right, bottom = rect
    got = ops.crop_resize(frames, rect)
    crop = frames[:, top:bottom + 1, left:right + 1].permute(0, 3, 1, 2).to(torch.float64)
    want = F_.interpolate(crop, size=(H, W), mode='bilinear', align_corners=False, antialias=False).permute(0, 2, 3, 1)
    diff = (got.to(torch.float64) - want).abs()
    assert float(diff.max()) <= 1.0 + 1e-9, float(diff.max())
    assert float(diff.mean()) < 0.35
    if rect == (0, 0, W - 1, H - 1):
        assert torch.equal(got, frames)
    # and on noise (worst case for interpolation): still within 1 LSB of the float64 result + its rounding
    noise = torch.randint(0, 256, (1, H, W, 3), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    got = ops.crop_resize(noise, rect)
    crop = noise[:, top:bottom + 1, left:right + 1].permute(0, 3, 1, 2).to(torch.float64)
    want = F_.interpolate(crop, size=(H, W), mode='bilinear', align_corners=False, antialias=False).permute(0, 2, 3, 1)
    assert float((got.to(torch.float64) - want).abs().max()) <= 1.0 + 1e-9
