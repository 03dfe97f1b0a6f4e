"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

The reference (`/root/reference/meshflowstabilizer.py`) does `import cv2` on line 1 and OpenCV is
not installed here, so a stub `cv2` module is injected first (only `FastFeatureDetector_create` is
touched by the constructor, mfs.py:99).  Every pure-NumPy method of the reference then runs as is:
`_get_adaptive_weights`, `_get_jacobi_method_input`, `_get_jacobi_method_output`,
`_get_stabilized_vertex_displacements`, `_get_vertex_x_y`, `_compute_stability_score`.

The reference itself never travels to the GPU box: only the vectors written here do (inputs +
the reference's outputs), together with this script.  Run from the repo root:

    python oracle/gen_golden.py            # needs /root/reference
    python oracle/gen_golden.py motion     # only the vertex-motion vectors (mfs.py:236-452)
    python oracle/gen_golden.py warp       # only the warp vectors (mfs.py:909-1108)

For the vertex-motion vectors the stub additionally provides `cv2.perspectiveTransform` and `cv2.medianBlur`
(the build's own restatements from oracle/), and `_get_matched_features_and_homography` (the FAST/LK/RANSAC
tracker, mfs.py:455-629) is replaced by a lookup into synthetic features; everything else -- the ellipse splat,
`statistics.median`, the dtype flow, the running sum -- is the reference's own code.

For the warp vectors (`warp_*.npz`) the REAL `_get_stabilized_frames_and_crop_boundaries` (mfs.py:909-1108) runs with
the four OpenCV calls it makes -- `cv2.findHomography` (mfs.py:1041-1042), `cv2.warpPerspective` (mfs.py:1052; the FULL
float64 bilinear warp of the mask image, not the non-zero-pattern shortcut), `cv2.perspectiveTransform` (mfs.py:1054)
and `cv2.remap` (mfs.py:1063-1069) -- served by the restatements in oracle/meshflow_oracle.py.  Everything else is the
reference's own NumPy: the int64 map templates (mfs.py:983-984), the row-major painter loop with `np.where` on the
mask's truthiness (mfs.py:1031-1061), the int64 -> float64 promotion of the maps, the four edge scans on the float64
maps (mfs.py:1075-1098) and the clip-level reduction (mfs.py:1103-1106).  So what stays unpinned in the warp half is
exactly those four cv2 kernels.
"""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')
REFERENCE_DIR = os.environ.get('MESHFLOW_REFERENCE_DIR', '/root/reference')


def import_reference():
    stub = types.ModuleType('cv2')
    stub.FastFeatureDetector_create = lambda *a, **k: None

    def perspective_transform(points, m):
        from oracle import meshflow_oracle as mo, motion_oracle as mt
        points = np.asarray(points)
        return mo.perspective_transform_f32(points, m) if points.dtype == np.float32 else mt.perspective_transform_f64(points, m)

    def median_blur(img, ksize):
        from oracle import motion_oracle as mt
        assert ksize == 3
        return mt.median_blur3_f32(img)

    def find_homography(src, dst, *args, **kwargs):
        from oracle import meshflow_oracle as mo
        assert not args and not kwargs                                # method 0, mfs.py:1041-1042
        return mo.find_homography_4pt(src, dst), None

    def warp_perspective(src, m, dsize, *args, **kwargs):
        from oracle import meshflow_oracle as mo
        assert not args and not kwargs                                # INTER_LINEAR, BORDER_CONSTANT 0, mfs.py:1052
        assert np.asarray(src).dtype == np.float64 and np.asarray(src).ndim == 2
        return mo.warp_perspective_f64_bilinear_np(src, m, dsize[0], dsize[1])

    def remap(src, map1, map2, interpolation, borderValue=(0, 0, 0)):
        from oracle import meshflow_oracle as mo
        assert interpolation == stub.INTER_LINEAR                      # mfs.py:1063-1069
        assert map1.dtype == np.float32 and map2.dtype == np.float32 and map1.shape == src.shape[:2] + (1,)
        return mo.remap_bilinear_u8c3(src, map1[..., 0], map2[..., 0], borderValue)

    stub.perspectiveTransform = perspective_transform
    stub.medianBlur = median_blur
    stub.findHomography = find_homography
    stub.warpPerspective = warp_perspective
    stub.remap = remap
    stub.INTER_LINEAR = 1
    sys.modules['cv2'] = stub
    sys.path.insert(0, REFERENCE_DIR)
    import meshflowstabilizer as mfs
    return mfs.MeshFlowStabilizer


def random_homographies(num_frames, seed):
    """Near-identity homographies with rotation/shear/scale/translation and small perspective rows
    (the perspective row is overwritten by the reference, mfs.py:816, and must not matter)."""
    from meshflow_amd import synthetic
    n = np.arange(num_frames * 9, dtype=np.int64).reshape(num_frames, 3, 3)
    g = synthetic.normal(n, seed)
    hom = np.tile(np.identity(3), (num_frames, 1, 1))
    hom[:, :2, :2] += 0.05 * g[:, :2, :2]
    hom[:, :2, 2] = 20.0 * g[:, :2, 2]
    hom[:, 2, :2] = 1e-4 * g[:, 2, :2]
    hom[-1] = np.identity(3)
    return hom


MOTION_CASES = (
    # name, W, H, R, C, ellipse rows, ellipse cols, frames, features per pair, seed
    ('motion_small', 640, 360, 8, 8, 5, 5, 6, (120, 180), 3),
    ('motion_1080p', 1920, 1080, 16, 16, 10, 10, 4, (500, 700), 0),
    ('motion_ragged', 100, 75, 3, 5, 3, 4, 5, (6, 14), 5),
)


def motion_inputs(W, H, R, C, F, per_pair, seed):
    """Features + homographies of one case, from the build's hash-based generator (regenerated in the tests)."""
    from meshflow_amd import synthetic
    n = np.arange((F - 1) * 9, dtype=np.int64).reshape(F - 1, 3, 3)
    g = synthetic.normal(n, seed * 16 + 13)
    hom = np.tile(np.identity(3), (F, 1, 1))
    hom[:-1, :2, :2] += 0.004 * g[:, :2, :2]
    hom[:-1, 0, 2] = 0.004 * W * g[:, 0, 2]
    hom[:-1, 1, 2] = 0.004 * H * g[:, 1, 2]
    hom[:-1, 2, :2] = 2e-6 * g[:, 2, :2]
    feats = synthetic.features(F, H, W, hom, seed=seed, per_pair=per_pair)
    return feats, hom


def motion_goldens(MFS):
    """(6) vertex-motion accumulation (mfs.py:236-452) -- see the module docstring for what is stubbed."""
    for name, W, H, R, C, er, ec, F, per_pair, seed in MOTION_CASES:
        feats, hom = motion_inputs(W, H, R, C, F, per_pair, seed)
        s = MFS(mesh_row_count=R, mesh_col_count=C, feature_ellipse_row_count=er, feature_ellipse_col_count=ec)
        frames = [np.broadcast_to(np.uint8(0), (H, W, 3)) for _ in range(F)]
        index = {id(f): i for i, f in enumerate(frames)}
        s._get_matched_features_and_homography = lambda early, late: (*feats[index[id(early)]], hom[index[id(early)]])
        disp, hom_out = s._get_unstabilized_vertex_displacements_and_homographies(F, frames)
        vel = np.stack([s._get_unstabilized_vertex_velocities(frames[t], frames[t + 1])[0] for t in range(F - 1)])
        assert vel.dtype == np.float32 and disp.dtype == np.float64 and np.array_equal(hom_out, hom)
        lx, ly = s._get_vertex_nearby_feature_residual_velocities(W, H, feats[0][0], feats[0][1], hom[0])
        counts = np.array([[len(v) for v in row] for row in lx], dtype=np.int32)
        flat_x = np.array([x for row in lx for v in row for x in v], dtype=np.float64)
        flat_y = np.array([y for row in ly for v in row for y in v], dtype=np.float64)
        assert all(isinstance(x, np.float64) for row in lx for v in row for x in v)
        np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), width=W, height=H, R=R, C=C, ell_rows=er, ell_cols=ec,
                            F=F, per_pair=np.array(per_pair), seed=seed, hom=hom,
                            offsets=np.cumsum([0] + [len(e) for e, _ in feats]),
                            early=np.concatenate([e.reshape(-1, 2) for e, _ in feats]),
                            late=np.concatenate([l.reshape(-1, 2) for _, l in feats]),
                            velocities=vel, displacements=disp, counts0=counts, lists0_x=flat_x, lists0_y=flat_y)


WARP_CASES = (
    # name, W, H, R, C, frames, frame kind, border (B, G, R), motion keywords, extra global shift (dx, dy) on the last frame
    ('warp_small', 64, 48, 3, 3, 3, 'noise', (0, 0, 255), dict(translation_sigma=1.5, field_sigma=0.6), None),
    ('warp_ragged', 101, 75, 5, 3, 2, 'noise', (0, 0, 255), dict(translation_sigma=2.0, field_sigma=1.0, jitter_sigma=0.3), None),   # R != C, W % 4 != 0
    ('warp_jitter', 96, 64, 4, 6, 3, 'noise', (17, 200, 3), dict(translation_sigma=2.0, field_sigma=1.0, jitter_sigma=2.0), None),   # strong iid vertex jitter
    ('warp_shift', 128, 72, 8, 8, 2, 'pattern', (0, 0, 255), dict(translation_sigma=0.5, field_sigma=0.3), (5.0, -3.0)),            # global shift of the last frame: wide uncovered bands
    ('warp_mesh16', 256, 144, 16, 16, 2, 'noise', (0, 0, 255), dict(translation_sigma=2.0, field_sigma=1.0, jitter_sigma=0.5), None),  # the default mesh
)


def warp_inputs(W, H, R, C, F, kind, motion_kw, shift, seed):
    """Frames + unstabilized / stabilized vertex displacements of one case (regenerated in the tests from the
    build's hash-based generator; the "stabilized" paths are a damped copy of the unstabilized ones, which is
    all the warp needs -- it only sees their difference, mfs.py:964-967)."""
    from meshflow_amd import synthetic
    frames = synthetic.frames_numpy(F, H, W, seed=seed, kind=kind)
    unstab, _ = synthetic.motion(F, R, C, seed=seed, **motion_kw)
    stab = 0.35 * unstab
    if shift is not None:
        stab = stab.copy()
        stab[-1] += np.asarray(shift)
    return frames, unstab, stab


def warp_goldens(MFS):
    """(7) the mesh warp and crop-boundary scan, mfs.py:909-1108 -- see the module docstring for what is stubbed."""
    for i, (name, W, H, R, C, F, kind, border, motion_kw, shift) in enumerate(WARP_CASES):
        frames, unstab, stab = warp_inputs(W, H, R, C, F, kind, motion_kw, shift, seed=40 + i)
        s = MFS(mesh_row_count=R, mesh_col_count=C, color_outside_image_area_bgr=border)
        out, bounds = s._get_stabilized_frames_and_crop_boundaries(F, list(frames), unstab, stab)
        assert len(out) == F and all(o.dtype == np.uint8 and o.shape == (H, W, 3) for o in out)
        assert all(isinstance(b, np.integer) for b in bounds)
        np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), width=W, height=H, R=R, C=C, F=F, kind=kind,
                            border=np.array(border), seed=40 + i, frames=frames, unstab=unstab, stab=stab,
                            out=np.stack(out), bounds=np.array(bounds, dtype=np.int64))
        print(name, 'bounds', tuple(int(b) for b in bounds), 'border pixels',
              int((np.stack(out) == np.array(border, dtype=np.uint8)).all(axis=-1).sum()))


def main():
    from meshflow_amd import synthetic
    MFS = import_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    if sys.argv[1:] == ['warp']:
        warp_goldens(MFS)
        print('warp vectors written to', GOLDEN)
        return
    if sys.argv[1:] == ['motion']:
        motion_goldens(MFS)
        print('vertex-motion vectors written to', GOLDEN)
        return

    # (1) coefficient setup, all four weight definitions  (mfs.py:713-841)
    out = {}
    for F, omega in ((12, 3), (48, 10), (300, 10), (300, 30)):
        hom = random_homographies(F, seed=100 + F + omega)
        for definition in range(4):
            s = MFS(temporal_smoothing_radius=omega)
            lam = s._get_adaptive_weights(F, 1920, 1080, definition, hom)
            off, on = s._get_jacobi_method_input(F, 1920, 1080, definition, hom)
            band = np.zeros((F, 2 * omega + 1))
            for d in range(-omega, omega + 1):
                idx = np.arange(max(0, -d), min(F, F - d))
                band[idx, d + omega] = off[idx, idx + d]
            outside = off.copy()
            for d in range(-omega, omega + 1):
                idx = np.arange(max(0, -d), min(F, F - d))
                outside[idx, idx + d] = 0
            assert not outside.any()
            key = f'F{F}_O{omega}_D{definition}'
            out[key + '_hom'] = hom
            out[key + '_lam'] = np.asarray(lam)
            out[key + '_band'] = band
            out[key + '_on'] = on
    np.savez_compressed(os.path.join(GOLDEN, 'coeffs.npz'), **out)

    # (2) small Jacobi problem, inputs + outputs, all four definitions  (mfs.py:632-710, 844-878)
    F, R, C, omega, iters = 48, 4, 4, 5, 25
    disp, hom = synthetic.motion(F, R, C, seed=7, jitter_sigma=0.5)
    frames = [np.zeros((360, 640, 3), dtype=np.uint8)]
    out = {'disp': disp, 'hom': hom, 'F': F, 'R': R, 'C': C, 'omega': omega, 'iters': iters,
           'width': 640, 'height': 360}
    for definition in range(4):
        s = MFS(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega,
                optimization_num_iterations=iters)
        res = s._get_stabilized_vertex_displacements(F, frames, definition, disp, hom)
        out[f'stab_D{definition}'] = np.ascontiguousarray(res)
    np.savez_compressed(os.path.join(GOLDEN, 'jacobi_small.npz'), **out)

    # (3) config-2 / config-3 sized problems: inputs come from the build's own generator (seeded,
    # hash based), the reference's outputs are stored for a subset of vertices.
    for name, F, R, C, omega, iters, nverts, definition in (
            ('jacobi_cfg2_subset', 300, 16, 16, 10, 100, 16, 0),
            ('jacobi_cfg2_high_subset', 300, 16, 16, 10, 100, 4, 2),
            ('jacobi_cfg3_subset', 600, 32, 32, 30, 200, 8, 0)):
        disp, hom = synthetic.motion(F, R, C, seed=0)
        s = MFS(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega,
                optimization_num_iterations=iters)
        off, on = s._get_jacobi_method_input(F, 1920, 1080, definition, hom)
        V = (R + 1) * (C + 1)
        verts = np.unique(np.linspace(0, V - 1, nverts).astype(np.int64))
        flat = disp.reshape(F, V, 2)
        res = np.stack([s._get_jacobi_method_output(off, on, flat[:, v], flat[:, v]) for v in verts], axis=1)
        np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), F=F, R=R, C=C, omega=omega, iters=iters,
                            definition=definition, seed=0, width=1920, height=1080, verts=verts,
                            inputs=flat[:, verts], outputs=res, hom=hom)

    # (4) vertex grids  (mfs.py:881-906)
    out = {}
    for W, H, R, C in ((1920, 1080, 16, 16), (1920, 1080, 32, 32), (3840, 2160, 16, 16), (640, 360, 16, 16),
                       (96, 64, 4, 4), (100, 75, 3, 5)):
        s = MFS(mesh_row_count=R, mesh_col_count=C)
        out[f'W{W}_H{H}_R{R}_C{C}'] = s._get_vertex_x_y(W, H)
    np.savez_compressed(os.path.join(GOLDEN, 'vertex_xy.npz'), **out)

    # (5) stability score  (mfs.py:1216-1259)
    out = {}
    for i, (F, R, C) in enumerate(((48, 4, 4), (120, 8, 8), (300, 16, 16))):
        disp, _ = synthetic.motion(F, R, C, seed=20 + i, jitter_sigma=0.3)
        s = MFS(mesh_row_count=R, mesh_col_count=C)
        out[f'disp{i}'] = disp
        out[f'score{i}'] = np.float64(s._compute_stability_score(F, disp))
    np.savez_compressed(os.path.join(GOLDEN, 'stability.npz'), **out)

    motion_goldens(MFS)
    warp_goldens(MFS)
    print('golden vectors written to', GOLDEN)


if __name__ == '__main__':
    main()
