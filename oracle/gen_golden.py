"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

The reference (`/root/reference/meshflowstabilizer.py`) does `import cv2` on line 1 and OpenCV is
not installed here, so a stub `cv2` module is injected first (only `FastFeatureDetector_create` is
touched by the constructor, mfs.py:99).  Every pure-NumPy method of the reference then runs as is:
`_get_adaptive_weights`, `_get_jacobi_method_input`, `_get_jacobi_method_output`,
`_get_stabilized_vertex_displacements`, `_get_vertex_x_y`, `_compute_stability_score`.

The reference itself never travels to the GPU box: only the vectors written here do (inputs +
the reference's outputs), together with this script.  Run from the repo root:

    python oracle/gen_golden.py            # needs /root/reference
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


def main():
    from meshflow_amd import synthetic
    MFS = import_reference()
    os.makedirs(GOLDEN, exist_ok=True)

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
    print('golden vectors written to', GOLDEN)


if __name__ == '__main__':
    main()
