"""`MeshFlowStabilizer.stabilize(input_path, output_path)` end to end: decode -> track -> (device) accumulate, smooth,
warp, crop -> scores -> encode.  OpenCV's part is played by tests/fake_cv2.py (OpenCV is not installed here), so this
checks the ORCHESTRATION and the device stages in between, not OpenCV.  Needs an MI355X."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture()
def cv2_stub(monkeypatch):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_cv2
    fake_cv2.VIDEOS.clear(); fake_cv2.WRITTEN.clear(); fake_cv2.MOTION.clear(); fake_cv2.EVENTS.clear()
    fake_cv2.READ_DELAY[0] = 0.0
    monkeypatch.setitem(sys.modules, 'cv2', fake_cv2.module())
    return fake_cv2


def _make_video(fake, path, F=10, H=96, W=128, claimed=None):
    from meshflow_amd import synthetic
    frames = [f.copy() for f in synthetic.frames_numpy(F, H, W, seed=4)]
    for t, f in enumerate(frames):
        fake.MOTION[id(f)] = (1.5 * np.sin(t), -1.0 * np.cos(2 * t))
    fake.VIDEOS[path] = dict(frames=frames, fps=30.0, fourcc=0x31637661, claimed=claimed)
    return frames


def test_stabilize_path_to_path(cv2_stub):
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    frames = _make_video(cv2_stub, 'in.m4v')
    s = MeshFlowStabilizer(mesh_row_count=4, mesh_col_count=4, mesh_outlier_subframe_row_count=2,
                           mesh_outlier_subframe_col_count=2, feature_ellipse_row_count=3, feature_ellipse_col_count=3,
                           temporal_smoothing_radius=3, optimization_num_iterations=10)
    ratio, distortion, stability = s.stabilize('in.m4v', 'out.m4v', MeshFlowStabilizer.ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL)
    rec = cv2_stub.WRITTEN['out.m4v']
    assert len(rec['frames']) == 10 and rec['fps'] == 30.0 and rec['fourcc'] == 0x31637661 and rec['size'] == (128, 96)
    assert isinstance(ratio, np.float32) and isinstance(distortion, np.float32) and 0.0 < float(stability) <= 1.0

    # the same result assembled by hand from the public pieces
    tracked = [s._get_matched_features_and_homography(a, b) for a, b in zip(frames[:-1], frames[1:])]
    assert all(e.dtype == np.float64 and e.shape[1:] == (1, 2) for e, _, _ in tracked)        # promoted, mfs.py:578
    hom = np.stack([h for _, _, h in tracked] + [np.identity(3)])
    disp, _ = s._get_unstabilized_vertex_displacements_from_features(10, 128, 96, [(e, l) for e, l, _ in tracked], hom)
    assert np.abs(disp[-1]).max() > 0.5
    _, bounds, stab, score, cropped = s.stabilize_clip(frames, disp, hom, crop=True, keep_uncropped=False)
    assert score == stability
    np.testing.assert_array_equal(np.stack(cropped), np.stack(rec['frames']))


def test_stabilize_short_video_raises_ioerror(cv2_stub):
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    _make_video(cv2_stub, 'short.m4v', F=5, claimed=8)
    with pytest.raises(IOError, match='did not have frame 5 of 8'):
        MeshFlowStabilizer(mesh_row_count=4, mesh_col_count=4).stabilize('short.m4v', 'out.m4v')


def test_stabilize_without_trackable_features_raises(cv2_stub):
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    _make_video(cv2_stub, 'in.m4v', F=4)
    s = MeshFlowStabilizer(mesh_row_count=4, mesh_col_count=4, homography_min_number_corresponding_features=10 ** 6)
    with pytest.raises(ValueError, match='features could be tracked'):
        s.stabilize('in.m4v', 'out.m4v')


def test_stabilize_honours_overridden_boundary_methods(cv2_stub):
    """mfs.py:150-159: `stabilize` reaches the hot path through the two private methods (and `_crop_frames`).  A subclass
    that overrides them -- the documented drop-in boundary -- must see its overrides called by the build's own `stabilize`,
    with the reference's argument lists, and the result must equal the un-overridden fast path."""
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    _make_video(cv2_stub, 'in.m4v')
    kw = dict(mesh_row_count=4, mesh_col_count=4, mesh_outlier_subframe_row_count=2, mesh_outlier_subframe_col_count=2,
              feature_ellipse_row_count=3, feature_ellipse_col_count=3, temporal_smoothing_radius=3, optimization_num_iterations=10)
    calls = []

    class Traced(MeshFlowStabilizer):
        def _get_stabilized_vertex_displacements(self, num_frames, unstabilized_frames, adaptive_weights_definition,
                                                 vertex_unstabilized_displacements_by_frame_index, homographies):
            calls.append(('jacobi', num_frames, adaptive_weights_definition, vertex_unstabilized_displacements_by_frame_index.shape))
            return super()._get_stabilized_vertex_displacements(num_frames, unstabilized_frames, adaptive_weights_definition,
                                                                vertex_unstabilized_displacements_by_frame_index, homographies)

        def _get_stabilized_frames_and_crop_boundaries(self, num_frames, unstabilized_frames, unstab, stab):
            calls.append(('warp', num_frames, len(unstabilized_frames), stab.shape))
            return super()._get_stabilized_frames_and_crop_boundaries(num_frames, unstabilized_frames, unstab, stab)

        def _crop_frames(self, uncropped_frames, crop_boundaries):
            calls.append(('crop', len(uncropped_frames), tuple(int(v) for v in crop_boundaries)))
            return super()._crop_frames(uncropped_frames, crop_boundaries)

    assert not MeshFlowStabilizer(**kw)._boundary_overridden() and Traced(**kw)._boundary_overridden()
    got = Traced(**kw).stabilize('in.m4v', 'traced.m4v', MeshFlowStabilizer.ADAPTIVE_WEIGHTS_DEFINITION_FLIPPED)
    assert [c[0] for c in calls] == ['jacobi', 'warp', 'crop']
    assert calls[0][1:] == (10, 1, (10, 5, 5, 2)) and calls[1][1:] == (10, 10, (10, 5, 5, 2)) and calls[2][1] == 10
    want = MeshFlowStabilizer(**kw).stabilize('in.m4v', 'plain.m4v', MeshFlowStabilizer.ADAPTIVE_WEIGHTS_DEFINITION_FLIPPED)
    assert got == want
    np.testing.assert_array_equal(np.stack(cv2_stub.WRITTEN['traced.m4v']['frames']), np.stack(cv2_stub.WRITTEN['plain.m4v']['frames']))
    # an instance-level patch counts as an override as well
    s = MeshFlowStabilizer(**kw)
    s._compute_stability_score = lambda num_frames, stab: 0.25
    assert s._boundary_overridden() and s.stabilize('in.m4v', 'patched.m4v')[2] == 0.25


KW = dict(mesh_row_count=4, mesh_col_count=4, mesh_outlier_subframe_row_count=2, mesh_outlier_subframe_col_count=2,
          feature_ellipse_row_count=3, feature_ellipse_col_count=3, temporal_smoothing_radius=3, optimization_num_iterations=10)


@pytest.mark.parametrize('F', [1 + 16, 40, 48])
def test_streamed_stabilize_equals_the_staged_sequence(cv2_stub, F):
    """the overlapped `stabilize` (streaming.py) against the stage-after-stage one (the reference's order,
    mfs.py:148-167): same return tuple, same encoded frames -- with a ragged last chunk, and with whole chunks only."""
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    _make_video(cv2_stub, 'in.m4v', F=F)
    staged = MeshFlowStabilizer(**KW)
    staged.overlap_video_io = False
    want = staged.stabilize('in.m4v', 'staged.m4v', MeshFlowStabilizer.ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL)
    got = MeshFlowStabilizer(**KW).stabilize('in.m4v', 'streamed.m4v', MeshFlowStabilizer.ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL)
    assert got == want and all(type(a) is type(b) for a, b in zip(got, want))
    a, b = cv2_stub.WRITTEN['streamed.m4v'], cv2_stub.WRITTEN['staged.m4v']
    assert len(a['frames']) == F and (a['fps'], a['fourcc'], a['size']) == (b['fps'], b['fourcc'], b['size'])
    np.testing.assert_array_equal(np.stack(a['frames']), np.stack(b['frames']))


def test_streamed_stabilize_overlaps_io_with_the_stages_around_it(cv2_stub, monkeypatch):
    """The order of events, not their speed: with a decoder that takes 2 ms per frame, the first chunk's tracking and upload must
    have started before the last frame is decoded; with downloads that finish 250 ms apart, the encoder must have been handed
    frames before the last chunk is back; and the frames reach the encoder in order."""
    import time
    from meshflow_amd import streaming as pipeline
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    F = 40
    frames = _make_video(cv2_stub, 'in.m4v', F=F)
    cv2_stub.READ_DELAY[0] = 0.002
    first_pair = id(frames[1])
    real_upload, real_flow = pipeline.ChunkedTransfer.upload, sys.modules['cv2'].calcOpticalFlowPyrLK

    def upload(self, clip, d_frames, i0, i1, after, k):
        cv2_stub.log('upload', k)
        return real_upload(self, clip, d_frames, i0, i1, after, k)

    def download(self, d_src, host_dst, after, k):
        def task():
            time.sleep(0.25 * k)          # (well above any one-time set-up cost of the first copy on a new stream)
            stream = self.out_streams[k % len(self.out_streams)]
            with torch.cuda.device(self.device), torch.cuda.stream(stream):
                stream.wait_event(after)
                torch.from_numpy(host_dst).copy_(d_src)
            cv2_stub.log('download', k)
        self.pending.append(self.out_pool.submit(task))

    def flow(early, late, points, nxt):
        cv2_stub.log('track', 0)
        return real_flow(early, late, points, nxt)

    monkeypatch.setattr(pipeline.ChunkedTransfer, 'upload', upload)
    monkeypatch.setattr(pipeline.ChunkedTransfer, 'download', download)
    monkeypatch.setattr(sys.modules['cv2'], 'calcOpticalFlowPyrLK', flow)
    MeshFlowStabilizer(**KW).stabilize('in.m4v', 'out.m4v')
    ev = list(cv2_stub.EVENTS)
    last_read = ev.index(('read', F - 1))
    assert ev.index(('upload', 0)) < last_read and ev.index(('upload', 1)) < last_read
    assert ev.index(('track', 0)) < last_read
    assert ev.index(('write', 0)) < ev.index(('download', 2))
    assert [i for kind, i in ev if kind == 'write'] == list(range(F))
    assert ev.index(('download', 0)) < ev.index(('write', 0))


def test_streamed_stabilize_is_bypassed_when_an_opencv_side_method_is_replaced(cv2_stub):
    """A subclass that replaces the decoder, tracker, score or encoder method keeps being called."""
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    _make_video(cv2_stub, 'in.m4v')
    seen = []

    class Custom(MeshFlowStabilizer):
        def _write_stabilized_video(self, output_path, num_frames, frames_per_second, codec, stabilized_frames):
            seen.append((output_path, num_frames, len(stabilized_frames)))

    Custom(**KW).stabilize('in.m4v', 'never.m4v')
    assert seen == [('never.m4v', 10, 10)] and 'never.m4v' not in cv2_stub.WRITTEN


def test_stabilize_clip_on_tiny_clips_and_dense_meshes():
    """`stabilize_clip(crop=True)` on clips of 2-9 frames of 2 x 2 to 101 x 50 pixels, radii up to 40 (far beyond the clip): paths within
    1e-9 of the NumPy oracle, frames / rectangle / cropped frames equal to the C oracle's warp of those paths and the NumPy crop of it.
    A one-frame clip has no velocity profile: np.fft refuses it inside the stability score, as in the reference (mfs.py:1241).  Meshes
    finer than a footprint (33 x 33 to 64 x 64 cells on frames of 65-200 pixels): the warp as the C oracle's."""
    import itertools
    import torch
    from meshflow_amd import ops
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    from oracle import clib, meshflow_oracle as mo
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(21)
    n = 0
    for F, (W, H), (R, C), omega in itertools.product((1, 2, 3, 9), ((2, 2), (4, 4), (5, 3), (8, 8), (33, 9), (64, 48), (101, 50)),
                                                      ((1, 1), (1, 3), (2, 2), (3, 1)), (1, 4, 40)):
        if C > W - 1 or R > H - 1:
            continue
        frames = [rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8) for _ in range(F)]
        vel = rng.normal(0, 0.6, size=(F, R + 1, C + 1, 2))
        vel[0] = 0
        disp = np.cumsum(vel, axis=0)
        hom = np.tile(np.eye(3), (F, 1, 1))
        hom[:, :2, 2] = rng.normal(0, 1.0, size=(F, 2))
        hom[-1] = np.eye(3)
        s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=5, device='cuda:0')
        if F == 1:
            with pytest.raises(ValueError):
                s.stabilize_clip(frames, disp, hom, crop=True)
            continue
        want_stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, omega, 5)
        _, first_crop, degenerate = clib.warp_clip(np.stack(frames), R, C, disp, want_stab, (0, 0, 255))
        empty = first_crop[:, 2].min() < first_crop[:, 0].max() or first_crop[:, 3].min() < first_crop[:, 1].max()
        if degenerate or empty:                              # (what cv2 would die of: the product refuses it, tested elsewhere)
            with pytest.raises(Exception):
                s.stabilize_clip(frames, disp, hom, crop=True)
            continue
        out, rect, stab, score, cropped = s.stabilize_clip(frames, disp, hom, crop=True)
        n += 1
        np.testing.assert_allclose(stab, want_stab, rtol=0, atol=1e-9)
        want, want_crop, _ = clib.warp_clip(np.stack(frames), R, C, disp, np.ascontiguousarray(stab), (0, 0, 255))
        want_rect = (int(want_crop[:, 0].max()), int(want_crop[:, 1].max()), int(want_crop[:, 2].min()), int(want_crop[:, 3].min()))
        assert tuple(int(v) for v in rect) == want_rect, (F, W, H, R, C, omega)
        assert np.array_equal(np.stack(out), want), (F, W, H, R, C, omega)
        assert np.array_equal(np.stack(cropped), np.stack(mo.crop_frames(list(want), want_rect))), (F, W, H, R, C, omega)
    assert n > 150
    for (W, H), (R, C), nfr in itertools.product(((65, 66), (80, 130), (128, 64), (200, 65)), ((33, 33), (48, 64), (64, 64), (64, 20), (7, 64)), (1, 2)):
        if C > W - 1 or R > H - 1:
            continue
        frames = rng.integers(0, 256, size=(nfr, H, W, 3), dtype=np.uint8)
        unstab = np.zeros((nfr, R + 1, C + 1, 2))
        stab = rng.normal(0, 1, size=(nfr, 1, 1, 2)) * 1.5 + rng.normal(0, 0.05, size=(nfr, R + 1, C + 1, 2))
        want, want_crop, degenerate = clib.warp_clip(frames, R, C, unstab, stab, (1, 2, 3))
        table = ops.cell_table(torch.from_numpy(unstab).to(dev), torch.from_numpy(stab).to(dev), W, H, R, C)
        out = ops.warp(torch.from_numpy(frames).to(dev), table, (1, 2, 3))
        assert int(table.status.item()) == degenerate
        if not degenerate:
            assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(table.crop.cpu().numpy(), want_crop), (W, H, R, C, nfr)


def test_drop_in_methods_accept_the_forms_numpy_code_hands_them():
    """The reference's methods take whatever NumPy takes (mfs.py:632, 909): views into a bigger stack, channel-reversed views
    (`frame[..., ::-1]`, the BGR / RGB idiom), Fortran-ordered frames, one 4-d array instead of a list, the transposed VIEW the
    reference's own path method returns (mfs.py:706-708), nested lists, float32 paths (promoted exactly, as NumPy would)."""
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    from oracle import clib
    F, H, W, R, C = 9, 48, 64, 3, 4
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=2, kind='noise', jitter_sigma=0.7)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=3, optimization_num_iterations=7, device='cuda:0')
    stab = s._get_stabilized_vertex_displacements(F, list(frames), 0, disp, hom)

    def expect(un, st):
        want, want_crop, bad = clib.warp_clip(frames, R, C, np.ascontiguousarray(un, dtype=np.float64), np.ascontiguousarray(st, dtype=np.float64), (0, 0, 255))
        assert bad == 0
        return want, (int(want_crop[:, 0].max()), int(want_crop[:, 1].max()), int(want_crop[:, 2].min()), int(want_crop[:, 3].min()))

    def check(fr, un, st):
        want, rect = expect(un, st)
        out, b = s._get_stabilized_frames_and_crop_boundaries(F, fr, un, st)
        assert np.array_equal(np.stack(out), want) and tuple(int(v) for v in b) == rect

    view = lambda a: np.moveaxis(np.ascontiguousarray(np.moveaxis(a, 0, 2)), 2, 0)      # non-contiguous, as mfs.py:706-708 returns it
    big = np.zeros((F, H + 6, W + 10, 3), np.uint8)
    big[:, 3:3 + H, 5:5 + W] = frames
    rev = np.ascontiguousarray(frames[..., ::-1])
    check(list(frames), disp, stab)
    check([big[i, 3:3 + H, 5:5 + W] for i in range(F)], disp, stab)
    check([rev[i][..., ::-1] for i in range(F)], disp, stab)
    check([np.asfortranarray(f) for f in frames], disp, stab)
    check(frames, disp, stab)
    check(list(frames), view(disp), view(stab))
    check(list(frames), disp.tolist(), stab.tolist())
    check(list(frames), disp.astype(np.float32), stab)
    assert np.array_equal(s._get_stabilized_vertex_displacements(F, list(frames), 0, view(disp), hom), stab)
    assert np.array_equal(s._get_stabilized_vertex_displacements(F, list(frames), 0, disp.tolist(), hom.tolist()), stab)


def test_stabilize_clip_from_several_host_threads():
    """The reference is re-entrant per instance (single-threaded NumPy, no shared state).  Here: four host threads with a stabilizer
    each (different clip shapes), then four threads sharing ONE stabilizer, twelve clips per thread through `stabilize_clip(crop=True)`
    -- the C pipeline serialises the calls on a device; every result equals the single-threaded one."""
    import threading
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    clips = []
    for seed in range(4):
        F, H, W, R, C = 24 + 4 * seed, 120, 160 + 32 * seed, 4, 4 + seed
        frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=seed, kind='noise', jitter_sigma=0.8)
        clips.append((list(frames), disp, hom, R, C))
    make = lambda R, C: MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=4, optimization_num_iterations=8, device='cuda:0')
    ref = []
    for fr, disp, hom, R, C in clips:
        out, rect, stab, score, cropped = make(R, C).stabilize_clip(fr, disp, hom, crop=True)
        ref.append((np.stack(out), tuple(int(v) for v in rect), stab, np.stack(cropped)))
    errors = []

    def worker(i, shared):
        fr, disp, hom, R, C = clips[i]
        s = shared if shared is not None else make(R, C)
        for _ in range(12):
            try:
                out, rect, stab, score, cropped = s.stabilize_clip(fr, disp, hom, crop=True)
                if not (np.array_equal(np.stack(out), ref[i][0]) and tuple(int(v) for v in rect) == ref[i][1]
                        and np.array_equal(stab, ref[i][2]) and np.array_equal(np.stack(cropped), ref[i][3])):
                    errors.append(('mismatch', i))
            except Exception as e:                                   # noqa: BLE001 -- reported below
                errors.append((type(e).__name__, str(e)[:100]))

    for shared in (None, make(clips[0][3], clips[0][4])):
        threads = [threading.Thread(target=worker, args=(i if shared is None else 0, shared)) for i in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors[:4]


def test_malformed_inputs_are_refused_with_python_errors():
    """The reference's error convention is Python exceptions (mfs.py:136-146, 205-208): a frame of another size, a grey or 4-channel
    frame, a clip shorter than its paths, paths of another mesh, homographies of another length, an empty clip, an unknown
    definition -> ValueError; frames that are not uint8 (float or 16-bit frames would take another cv2.remap code path) -> TypeError.
    Read-only frames and the same frame object six times are fine.  The path method looks at the first frame's SHAPE only."""
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    F, H, W, R, C = 6, 48, 64, 2, 2
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=3, kind='noise')
    fl = list(frames)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=3, optimization_num_iterations=5, device='cuda:0')
    for bad in (fl[:3] + [np.zeros((40, 64, 3), np.uint8)] + fl[4:], fl[:3] + [np.zeros((48, 64), np.uint8)] + fl[4:],
                [np.zeros((48, 64, 4), np.uint8)] * F, fl[:4], []):
        with pytest.raises(ValueError):
            s.stabilize_clip(bad, disp, hom)
    with pytest.raises(ValueError):
        s.stabilize_clip(fl, disp[:, :2], hom)
    with pytest.raises(ValueError):
        s.stabilize_clip(fl, disp, hom[:3])
    with pytest.raises(ValueError):
        s.stabilize_clip(fl, disp, hom, adaptive_weights_definition=7)
    for bad in ([f.astype(np.float32) for f in fl], [f.astype(np.uint16) for f in fl], frames.astype(np.float64)):
        with pytest.raises(TypeError):
            s.stabilize_clip(bad, disp, hom)
        with pytest.raises(TypeError):
            s._get_stabilized_frames_and_crop_boundaries(F, bad, disp, disp)
    ref = s.stabilize_clip(fl, disp, hom, crop=True)
    ro = s.stabilize_clip([np.frombuffer(f.tobytes(), np.uint8).reshape(f.shape) for f in fl], disp, hom, crop=True)
    assert np.array_equal(np.stack(ro[0]), np.stack(ref[0])) and np.array_equal(np.stack(ro[4]), np.stack(ref[4]))
    same = s.stabilize_clip([fl[0]] * F, disp, hom, crop=True)
    assert len(same[0]) == F
    stab = s._get_stabilized_vertex_displacements(F, [f.astype(np.float32) for f in fl], 0, disp, hom)      # (shape only: mfs.py:679)
    assert np.array_equal(stab, ref[2])
