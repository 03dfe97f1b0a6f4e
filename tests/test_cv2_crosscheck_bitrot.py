"""tests/test_cv2_crosscheck.py is the door to "parity green" -- one run anywhere a real OpenCV exists pins the five restated OpenCV
calls -- and it has never executed (no cv2 in the build image or on the GPU box: it skips at import).  This module keeps it from
rotting: it imports that file with a STAND-IN `cv2` whose entry points are the oracle's own restatements (the stub oracle/gen_golden.py
runs the reference under, plus cv2.resize) and runs every test in it, so that a renamed oracle function, a changed signature or a
default that moved (the 4-point solver's did in round 4) breaks HERE, on the CPU, and not on the day a real cv2 turns up.  Passing
proves nothing about OpenCV -- the stand-in IS the oracle; it proves the cross-check still calls the current oracle API correctly."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _stand_in_cv2():
    from oracle import meshflow_oracle as mo, motion_oracle as mt
    m = types.ModuleType('cv2')
    m.__stand_in__ = True
    m.INTER_LINEAR = 1
    m.FastFeatureDetector_create = lambda *a, **k: None

    def find_homography(src, dst, *args, **kwargs):
        return mo.find_homography_4pt(src, dst), None

    def warp_perspective(src, h, dsize, *args, **kwargs):
        return mo.warp_perspective_f64_bilinear_np(np.asarray(src, dtype=np.float64), h, dsize[0], dsize[1])

    def perspective_transform(points, h):
        points = np.asarray(points)
        return mo.perspective_transform_f32(points, h) if points.dtype == np.float32 else mt.perspective_transform_f64(points, h)

    def remap(src, map1, map2, interpolation, borderValue=(0, 0, 0)):
        assert interpolation == m.INTER_LINEAR
        return mo.remap_bilinear_u8c3(src, map1[..., 0], map2[..., 0], borderValue)

    m.findHomography, m.warpPerspective, m.perspectiveTransform, m.remap = find_homography, warp_perspective, perspective_transform, remap
    m.resize = lambda src, dsize, *a, **k: mo.resize_linear_u8(src, dsize[0], dsize[1])
    m.medianBlur = lambda img, ksize: mt.median_blur3_f32(img)
    return m


@pytest.fixture(scope='module')
def crosscheck():
    """tests/test_cv2_crosscheck.py imported under the stand-in (under another module name; sys.modules is put back afterwards)."""
    saved = {k: sys.modules.get(k) for k in ('cv2', 'meshflowstabilizer')}
    sys.modules['cv2'] = _stand_in_cv2()
    sys.modules.pop('meshflowstabilizer', None)
    try:
        spec = importlib.util.spec_from_file_location('_cv2_crosscheck_under_stand_in', os.path.join(HERE, 'test_cv2_crosscheck.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert getattr(mod.cv2, '__stand_in__', False)
        yield mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _tests_of(mod):
    return sorted(n for n in dir(mod) if n.startswith('test_') and callable(getattr(mod, n)))


def test_the_crosscheck_module_still_collects(crosscheck):
    names = _tests_of(crosscheck)
    assert len(names) >= 14 and 'test_find_homography_4pt' in names and 'test_whole_warp_against_reference_loop' in names


@pytest.mark.parametrize('name', ['test_find_homography_4pt', 'test_perspective_transform', 'test_warp_perspective_mask_pattern',
                                  'test_remap_bilinear_constant_border', 'test_resize_linear', 'test_whole_warp_against_reference_loop',
                                  'test_perspective_transform_float64_points', 'test_median_blur_3x3_float32',
                                  'test_vertex_velocities_against_reference_code_with_real_cv2', 'test_cfg2_find_homography_both_directions',
                                  'test_cfg2_resize_of_the_real_crop_rectangle'])
def test_crosscheck_runs_against_the_current_oracle_api(crosscheck, name):
    getattr(crosscheck, name)()


def test_crosscheck_full_frame_tests_run(crosscheck, monkeypatch):
    """The two 1920x1080 tests, on one cell instead of six (the stand-in's warpPerspective is a NumPy full-frame pass)."""
    cells = crosscheck._cfg2_cells
    monkeypatch.setattr(crosscheck, '_cfg2_cells', lambda n_cells=24, seed=11: cells(n_cells=min(n_cells, 1), seed=seed))
    crosscheck.test_cfg2_warp_perspective_mask_and_perspective_transform()
    crosscheck.test_cfg2_remap_full_frame()


def test_crosscheck_golden_test_runs(crosscheck):
    """The reference's own warp under the stand-in against a committed golden (what oracle/gen_golden.py made it from): byte-equal here."""
    crosscheck.test_reference_with_real_cv2_reproduces_the_committed_goldens('warp_small')


def _import_crosscheck_under(cv2_module, name):
    saved = sys.modules.get('cv2')
    sys.modules['cv2'] = cv2_module
    try:
        spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, 'test_cv2_crosscheck.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        if saved is None:
            sys.modules.pop('cv2', None)
        else:
            sys.modules['cv2'] = saved


@pytest.mark.parametrize('version,bit_exact', [('4.5.5', True), ('4.10.0', True), ('4.11.0', False), ('4.12.0-dev', False), ('3.4.16', False)])
def test_crosscheck_is_version_aware(version, bit_exact):
    """Inside the modelled OpenCV range (4.5 ... 4.10: fixed-point remap / warpPerspective / resize) a single differing byte fails the
    cross-check; outside it (>= 4.11: float-weight kernels) a last-bit difference passes -- and is REPORTED -- while 2 LSB still fails."""
    fake = _stand_in_cv2()
    fake.__version__ = version
    exact_remap = fake.remap

    def off_by(n):
        def remap(src, map1, map2, interpolation, borderValue=(0, 0, 0)):
            out = exact_remap(src, map1, map2, interpolation, borderValue).astype(np.int64)
            out[3::7, 5::11] += np.where(out[3::7, 5::11] < 200, n, -n)
            return out.astype(np.uint8)
        return remap

    mod = _import_crosscheck_under(fake, f'_cv2_crosscheck_version_{version.replace(".", "_").replace("-", "_")}')
    assert mod.BIT_EXACT is bit_exact and mod._REPORT['cv2_version'] == version
    mod.test_remap_bilinear_constant_border()                         # the exact stand-in passes in either mode
    fake.remap = off_by(1)
    if bit_exact:
        with pytest.raises(AssertionError):
            mod.test_remap_bilinear_constant_border()
    else:
        mod.test_remap_bilinear_constant_border()
        last = mod._REPORT['checks'][-1]
        assert last['mismatching'] > 0 and last['max_abs'] == 1        # reported as information
    fake.remap = off_by(2)
    with pytest.raises(AssertionError):
        mod.test_remap_bilinear_constant_border()
