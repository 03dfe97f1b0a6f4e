"""`MeshFlowStabilizer`: the reference's class surface with its two hot methods on an MI355X.

Mirrors /root/reference/meshflowstabilizer.py (`mfs.py`): constructor keywords and defaults
(mfs.py:43-49), the four enum values and two constants (mfs.py:32-40), `stabilize(input_path,
output_path, adaptive_weights_definition)` and its `(cropping_ratio, distortion_score,
stability_score)` return (mfs.py:102-169), and the two private methods that form the drop-in boundary:

    _get_stabilized_vertex_displacements          mfs.py:632-710    Jacobi sweep      -> HIP kernel 1
    _get_stabilized_frames_and_crop_boundaries    mfs.py:909-1108   mesh warp + crop  -> HIP kernels 2a/2b

Both take and return NumPy arrays exactly like the reference (host buffers in, host buffers out); the
`*_device` variants keep everything in HBM and are what `stabilize_clip` and bench.py use.  Video
decode/encode, the FAST/LK/RANSAC front-end and the two feature-based metrics are outside this path
(they need OpenCV); `stabilize()` needs `cv2` for them and says so when it is missing.
"""
import numpy as np

from . import host

try:  # progress bars are optional
    import tqdm as _tqdm  # noqa: F401
except Exception:  # pragma: no cover
    _tqdm = None


def _torch_device_of(self):
    """The torch device of a stabilizer(-like) object: its `device` attribute, None / absent / 'cuda' = the CURRENT HIP device."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('no MI355X visible: the meshflow_amd hot path has no CPU fallback')
    name = getattr(self, 'device', None)
    dev = torch.device(name if name is not None else 'cuda')
    if dev.index is None:                 # 'cuda' without an index means the caller's CURRENT device, not device 0
        dev = torch.device(dev.type, torch.cuda.current_device())
    return dev


def _warp_host_c(self, clip, unstab, stab, crop=False, keep_uncropped=True):
    """Host frames in -> host frames out through the C ABI's own chunked pipeline (csrc/hostpipe.hip:
    `mf_warp_u8c3_host_frames`, upload / kernel / download threads below Python, GIL released for the whole call).
    Returns (stabilized frames (F, H, W, 3) uint8 array, clip-level crop bounds as np.int64 (left, top, right,
    bottom), mfs.py:1103-1106).  crop=True: `mf_warp_crop_u8c3_host_frames` -- the same pipeline followed by `_crop_frames`
    (mfs.py:159, 1111-1157) on the device; returns (stabilized frames or None when keep_uncropped is False, bounds,
    cropped + resized frames (F, H, W, 3))."""
    import ctypes
    import torch
    from . import _lib, pipeline
    dev = self._torch_device()
    n, H, W = clip.num_frames, clip.height, clip.width
    frames = [clip.array[i] for i in range(n)] if clip.array is not None else clip.frames      # (validated by HostClip, every one of them)
    fb = H * W * 3
    want_out = keep_uncropped or not crop
    out = np.empty((n, H, W, 3), dtype=np.uint8) if want_out else None
    pin = (ctypes.c_void_p * n)(*[f.ctypes.data for f in frames])
    pout = (ctypes.c_void_p * n)(*[out.ctypes.data + i * fb for i in range(n)]) if want_out else None
    per_frame = np.empty((n, 4), dtype=np.int32)
    border = (ctypes.c_uint8 * 3)(*[int(max(0, min(255, round(float(c))))) for c in self.color_outside_image_area_bgr])
    args = (unstab.ctypes.data_as(ctypes.c_void_p), stab.ctypes.data_as(ctypes.c_void_p), n, W, H,
            self.mesh_row_count, self.mesh_col_count, border, per_frame.ctypes.data_as(ctypes.c_void_p))
    # the library works on the calling thread's current HIP device: scope it like every other path here does (and leave the
    # caller's current device as it was)
    with torch.cuda.device(dev):
        if not crop:
            _lib.check(_lib.lib.mf_warp_u8c3_host_frames(pin, pout, *args, None))
            bounds = (np.int64(per_frame[:, 0].max()), np.int64(per_frame[:, 1].max()),
                      np.int64(per_frame[:, 2].min()), np.int64(per_frame[:, 3].min()))
            return out, bounds
        cropped = np.empty((n, H, W, 3), dtype=np.uint8)
        pcrop = (ctypes.c_void_p * n)(*[cropped.ctypes.data + i * fb for i in range(n)])
        rect = (ctypes.c_int32 * 4)()
        _lib.check(_lib.lib.mf_warp_crop_u8c3_host_frames(pin, pout, pcrop, *args, rect, None))
    return out, tuple(np.int64(v) for v in rect), cropped


class DegenerateMeshError(ValueError):
    """A resident clip had mesh cells without a homography (cv2.findHomography would return None and the reference dies inside cv2,
    mfs.py:1041-1042); its frames are undefined.  `clip_serial`: the clip's serial number -- `MeshFlowStabilizer.resident_serial` right
    after the `stabilize_resident` call that issued it; `cells`: how many cells.  A ValueError, as before."""

    def __init__(self, cells, clip_serial):
        super().__init__(f'{cells} degenerate mesh cell(s) in resident clip #{clip_serial}: no homography exists '
                         '(cv2.findHomography would return None); its frames are undefined')
        self.cells, self.clip_serial = cells, clip_serial


class MeshFlowStabilizer:
    ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL = 0
    ADAPTIVE_WEIGHTS_DEFINITION_FLIPPED = 1
    ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_HIGH = 2
    ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_LOW = 3

    ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_HIGH_VALUE = 100
    ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_LOW_VALUE = 1

    def __init__(self, mesh_row_count=16, mesh_col_count=16,
                 mesh_outlier_subframe_row_count=4, mesh_outlier_subframe_col_count=4,
                 feature_ellipse_row_count=10, feature_ellipse_col_count=10,
                 homography_min_number_corresponding_features=4,
                 temporal_smoothing_radius=10, optimization_num_iterations=100,
                 color_outside_image_area_bgr=(0, 0, 255),
                 visualize=False, device=None):
        self.mesh_col_count = mesh_col_count
        self.mesh_row_count = mesh_row_count
        self.mesh_outlier_subframe_row_count = mesh_outlier_subframe_row_count
        self.mesh_outlier_subframe_col_count = mesh_outlier_subframe_col_count
        self.feature_ellipse_row_count = feature_ellipse_row_count
        self.feature_ellipse_col_count = feature_ellipse_col_count
        self.homography_min_number_corresponding_features = homography_min_number_corresponding_features
        self.temporal_smoothing_radius = temporal_smoothing_radius
        self.optimization_num_iterations = optimization_num_iterations
        self.color_outside_image_area_bgr = color_outside_image_area_bgr
        self.visualize = visualize
        self.device = device          # extra keyword: torch device string; None = current HIP device

    # ------------------------------------------------------------------------------------------
    # public API
    # ------------------------------------------------------------------------------------------

    @staticmethod
    def _check_definition(adaptive_weights_definition):
        if adaptive_weights_definition not in host.VALID_DEFINITIONS:          # mfs.py:136-146
            raise ValueError(
                'Invalid value for `adaptive_weights_definition`. Expecting value of '
                '`MeshFlowStabilizer.ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL`, '
                '`MeshFlowStabilizer.ADAPTIVE_WEIGHTS_DEFINITION_FLIPPED`, '
                '`MeshFlowStabilizer.ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_HIGH`, or'
                '`MeshFlowStabilizer.ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_LOW`.')

    def stabilize(self, input_path, output_path, adaptive_weights_definition=ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL):
        """Same contract as mfs.py:102-169: reads `input_path`, writes the stabilized video to `output_path`, returns
        (cropping_ratio, distortion_score, stability_score).  Video I/O, the FAST/LK/RANSAC tracker and the two
        feature-based scores are OpenCV calls in the reference and stay OpenCV calls here (`frontend_cv2.py`, needs
        `cv2`); everything between the tracker and the encoder runs on the MI355X."""
        self._check_definition(adaptive_weights_definition)
        from . import frontend_cv2
        cv2 = frontend_cv2.require_cv2()
        if self.overlap_video_io and not self._boundary_overridden(self._BOUNDARY_METHODS + self._OPENCV_METHODS):
            # nothing replaced: decode || track || upload, then the device path, then download || encode || scores
            from . import streaming
            return streaming.stabilize_streamed(self, cv2, input_path, output_path, adaptive_weights_definition)
        unstabilized_frames, num_frames, frames_per_second, codec = self._get_unstabilized_frames_and_video_features(input_path)
        disp, homographies = self._get_unstabilized_vertex_displacements_and_homographies(num_frames, unstabilized_frames)
        if self._boundary_overridden():
            # A subclass (or an instance attribute) replaces one of the reference's boundary methods: keep the reference's
            # own call sequence (mfs.py:150-162) so that the replacement is honoured, at the price of one PCIe round trip
            # per method instead of one for the whole path.
            stab = self._get_stabilized_vertex_displacements(
                num_frames, unstabilized_frames, adaptive_weights_definition, disp, homographies)               # mfs.py:150
            stabilized_frames, crop_boundaries = self._get_stabilized_frames_and_crop_boundaries(
                num_frames, unstabilized_frames, disp, stab)                                                     # mfs.py:154
            cropped_frames = self._crop_frames(stabilized_frames, crop_boundaries)                               # mfs.py:159
            stability_score = self._compute_stability_score(num_frames, stab)                                    # mfs.py:162
        else:
            _, _, stab, stability_score, cropped_frames = self.stabilize_clip(
                unstabilized_frames, disp, homographies, adaptive_weights_definition, crop=True, keep_uncropped=False)
        cropping_ratio, distortion_score = self._compute_cropping_ratio_and_distortion_score(
            num_frames, unstabilized_frames, cropped_frames)
        self._write_stabilized_video(output_path, num_frames, frames_per_second, codec, cropped_frames)
        if self.visualize:
            frontend_cv2.show_loop(cv2, frames_per_second, unstabilized_frames, cropped_frames)
        return (cropping_ratio, distortion_score, stability_score)

    _BOUNDARY_METHODS = ('_get_stabilized_vertex_displacements', '_get_stabilized_frames_and_crop_boundaries',
                         '_crop_frames', '_compute_stability_score')

    _OPENCV_METHODS = ('_get_unstabilized_frames_and_video_features', '_get_unstabilized_vertex_displacements_and_homographies',
                       '_get_matched_features_and_homography', '_compute_cropping_ratio_and_distortion_score',
                       '_write_stabilized_video')
    overlap_video_io = True      # False: `stabilize` runs its stages one after the other, like the reference

    def _boundary_overridden(self, names=None):
        """True when one of the reference's boundary methods (called at mfs.py:150-162) is not this class's
        own implementation: overridden in a subclass or patched on the instance."""
        for name in names or self._BOUNDARY_METHODS:
            if name in vars(self) or getattr(type(self), name) is not getattr(MeshFlowStabilizer, name):
                return True
        return False

    # ---- the reference's OpenCV-side helpers, same names, delegating to cv2 (frontend_cv2.py) ----

    def _tracker(self):
        from . import frontend_cv2
        return frontend_cv2.Tracker(frontend_cv2.require_cv2(), self.mesh_outlier_subframe_row_count,
                                    self.mesh_outlier_subframe_col_count,
                                    self.homography_min_number_corresponding_features)

    def _get_unstabilized_frames_and_video_features(self, input_path):
        """mfs.py:172-213."""
        from . import frontend_cv2
        return frontend_cv2.read_video(frontend_cv2.require_cv2(), input_path)

    def _get_matched_features_and_homography(self, early_frame, late_frame):
        """mfs.py:455-528."""
        return self._tracker().track_pair(early_frame, late_frame)

    def _get_unstabilized_vertex_displacements_and_homographies(self, num_frames, unstabilized_frames):
        """mfs.py:236-284: the tracker on every adjacent frame pair (cv2, host threads), then the accumulation of
        their features into vertex displacements on the device."""
        tracked = self._tracker().track_pairs(unstabilized_frames[:-1], unstabilized_frames[1:])
        homographies = np.empty((num_frames, 3, 3))
        homographies[-1] = np.identity(3)                                                   # mfs.py:274
        for t, (_, _, h) in enumerate(tracked):
            if h is None:      # the reference hands None to cv2.perspectiveTransform (mfs.py:325) and dies in cv2
                raise ValueError(f'fewer than {self.homography_min_number_corresponding_features} features could be '
                                 f'tracked from frame {t} to frame {t + 1}')
            homographies[t] = h
        frame_height, frame_width = unstabilized_frames[0].shape[:2]
        return self._get_unstabilized_vertex_displacements_from_features(
            num_frames, frame_width, frame_height, [(e, l) for e, l, _ in tracked], homographies)

    def _compute_cropping_ratio_and_distortion_score(self, num_frames, unstabilized_frames, cropped_frames):
        """mfs.py:1160-1212."""
        from . import frontend_cv2
        return frontend_cv2.cropping_and_distortion(self._tracker(), unstabilized_frames[:num_frames],
                                                    cropped_frames[:num_frames])

    def _write_stabilized_video(self, output_path, num_frames, frames_per_second, codec, stabilized_frames):
        """mfs.py:1290-1322."""
        from . import frontend_cv2
        frontend_cv2.write_video(frontend_cv2.require_cv2(), output_path, frames_per_second, codec,
                                 stabilized_frames[:num_frames])

    def stabilize_clip(self, unstabilized_frames, vertex_unstabilized_displacements_by_frame_index, homographies,
                       adaptive_weights_definition=ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL, crop=False,
                       keep_uncropped=True, chunk_frames=16, io_threads=3):
        """The hot path of `stabilize` (mfs.py:150-159, 162) on in-memory inputs, with ONE host->device and ONE
        device->host pass over the frames (the two private methods below each pay their own, like any drop-in
        for NumPy-in / NumPy-out methods must).  The passes are chunked and overlapped with each other and with
        the kernels below Python (csrc/hostpipe.hip): each 16-frame chunk is warped as soon as it has landed and
        comes back while later chunks are still going up.  (`chunk_frames` / `io_threads` are kept for callers of
        earlier versions; the C pipeline has its own, MF_PIPE_* in the environment.)

        Returns (stabilized_frames list, crop_boundaries, vertex_stabilized_displacements, stability_score)
        and, with crop=True, a fifth item: the cropped + resized frames (`_crop_frames`, mfs.py:159), produced
        on the device from the stabilized frames in the same pipeline once the clip-level rectangle is known
        (`mf_warp_crop_u8c3_host_frames`); keep_uncropped=False then never brings the uncropped stabilized frames back
        (the reference only uses the cropped ones afterwards) and returns None for them."""
        import torch
        from . import pipeline
        self._check_definition(adaptive_weights_definition)
        if chunk_frames != 16 or io_threads != 3:
            import warnings
            warnings.warn('stabilize_clip: chunk_frames / io_threads are ignored since the frames travel through the C pipeline '
                          '(csrc/hostpipe.hip); tune it with MF_PIPE_CHUNK / MF_PIPE_UP / MF_PIPE_DOWN in the environment', DeprecationWarning, stacklevel=2)
        num_frames = len(unstabilized_frames)
        dev = self._torch_device()
        unstab = np.ascontiguousarray(vertex_unstabilized_displacements_by_frame_index, dtype=np.float64)
        self._check_mesh_shape(unstab, num_frames)
        clip = pipeline.HostClip(unstabilized_frames, num_frames)
        H, W = clip.height, clip.width
        # the frames travel through the C ABI's own pipeline (no Python threads, GIL released): up in chunks, warped as they land,
        # and -- crop=True -- cropped + resized on the device once the clip-level rectangle is known, only then back
        d_stab = self._stabilized_vertex_displacements_device(torch.from_numpy(unstab).to(dev), W, H,
                                                              adaptive_weights_definition, homographies)
        stab = d_stab.cpu().numpy()
        # the stability score (mfs.py:162: np.fft over every vertex path, 20 ms at config 3) on a host thread BESIDE the frames' trip over
        # PCIe (the C call releases the GIL): same function, same bits, off the critical path
        import threading
        score = {}

        def scoring():
            try:
                score['value'] = self._compute_stability_score(num_frames, stab)
            except BaseException as e:          # re-raised on the calling thread
                score['error'] = e
        scorer = threading.Thread(target=scoring, name='mf-score')
        scorer.start()
        try:
            res = _warp_host_c(self, clip, unstab, stab, crop=crop, keep_uncropped=keep_uncropped)
        finally:
            scorer.join()
        if 'error' in score:
            raise score['error']
        if not crop:
            out_host, bounds = res
            return list(out_host), bounds, stab, score['value']
        out_host, bounds, cropped_host = res
        return (list(out_host) if out_host is not None else None), bounds, stab, score['value'], list(cropped_host)

    # ------------------------------------------------------------------------------------------
    # drop-in boundary, host buffers (same signatures as the reference)
    # ------------------------------------------------------------------------------------------

    def _torch_device(self):
        return _torch_device_of(self)

    def _get_stabilized_vertex_displacements(self, num_frames, unstabilized_frames, adaptive_weights_definition,
                                             vertex_unstabilized_displacements_by_frame_index, homographies):
        """mfs.py:632-710.  float64 (F, R+1, C+1, 2) in, float64 (F, R+1, C+1, 2) out."""
        import torch
        disp = np.ascontiguousarray(vertex_unstabilized_displacements_by_frame_index, dtype=np.float64)
        self._check_mesh_shape(disp, num_frames)
        frame_height, frame_width = unstabilized_frames[0].shape[:2]                       # mfs.py:679
        dev = self._torch_device()
        d_disp = torch.from_numpy(disp).to(dev)
        d_stab = self._stabilized_vertex_displacements_device(d_disp, frame_width, frame_height,
                                                              adaptive_weights_definition, homographies)
        return d_stab.cpu().numpy()

    def _get_stabilized_frames_and_crop_boundaries(self, num_frames, unstabilized_frames,
                                                   vertex_unstabilized_displacements_by_frame_index,
                                                   vertex_stabilized_displacements_by_frame_index,
                                                   chunk_frames=16, io_threads=3):
        """mfs.py:909-1108.  Returns (list of F uint8 (H, W, 3) arrays, (left, top, right, bottom)).
        (`chunk_frames` / `io_threads` are kept for callers of earlier versions; the C pipeline has its own.)"""
        from . import pipeline
        unstab = np.ascontiguousarray(vertex_unstabilized_displacements_by_frame_index, dtype=np.float64)
        stab = np.ascontiguousarray(vertex_stabilized_displacements_by_frame_index, dtype=np.float64)
        self._check_mesh_shape(unstab, num_frames)
        self._check_mesh_shape(stab, num_frames)
        clip = pipeline.HostClip(unstabilized_frames, num_frames)
        out_host, bounds = _warp_host_c(self, clip, unstab, stab)
        return list(out_host), bounds

    def _get_unstabilized_vertex_displacements_from_features(self, num_frames, frame_width, frame_height,
                                                             features_by_pair, homographies):
        """mfs.py:236-284 with the tracker's outputs injected: `features_by_pair[t]` = (early_features,
        late_features) of frames t, t+1 as `_get_matched_features_and_homography` returns them (mfs.py:455-528),
        `homographies` (F, 3, 3) with the identity last (mfs.py:274).  Returns the reference's tuple
        (vertex_unstabilized_displacements_by_frame_index float64 (F, R+1, C+1, 2), homographies)."""
        disp, _ = self._vertex_motion_from_features(num_frames, frame_width, frame_height, features_by_pair, homographies)
        return disp, np.asarray(homographies, dtype=np.float64)

    def _get_unstabilized_vertex_velocities_from_features(self, frame_width, frame_height, early_features,
                                                          late_features, early_to_late_homography):
        """mfs.py:287-362 after the tracker call at :316: float32 (R+1, C+1, 2) velocities of one frame pair."""
        hom = np.asarray(early_to_late_homography, dtype=np.float64).reshape(1, 3, 3)
        _, vel = self._vertex_motion_from_features(2, frame_width, frame_height, [(early_features, late_features)], hom)
        return vel[0]

    def _vertex_motion_from_features(self, num_frames, frame_width, frame_height, features_by_pair, homographies):
        import torch
        from . import ops
        if len(features_by_pair) != num_frames - 1:
            raise ValueError('features_by_pair must hold num_frames - 1 (early, late) pairs')
        hom = np.ascontiguousarray(np.asarray(homographies, dtype=np.float64)[:num_frames - 1]).reshape(-1, 3, 3)
        if hom.shape[0] != num_frames - 1:
            raise ValueError('homographies must hold at least num_frames - 1 matrices')
        early, late, offsets, kmax = host.pack_features(features_by_pair)
        dev = self._torch_device()
        d_disp, d_vel, status = ops.vertex_motion(
            torch.from_numpy(early).to(dev), torch.from_numpy(late).to(dev), torch.from_numpy(offsets).to(dev),
            torch.from_numpy(hom).to(dev), kmax, frame_width, frame_height, self.mesh_row_count, self.mesh_col_count,
            self.feature_ellipse_row_count, self.feature_ellipse_col_count)
        ops.vertex_motion_check(status)
        return d_disp.cpu().numpy(), d_vel.cpu().numpy()

    def _crop_frames(self, uncropped_frames, crop_boundaries, chunk_frames=16, io_threads=3):
        """mfs.py:1111-1157: crop to the inclusive bounds and resize back to (W, H) (cv2.resize, INTER_LINEAR).
        Host frames in, host frames out through the C ABI's ring of chunk buffers (`mf_crop_resize_u8c3_host_frames`,
        csrc/hostpipe.hip): upload, resize and download of the chunks all overlap, below Python.
        (`chunk_frames` / `io_threads` are kept for callers of earlier versions; the C pipeline has its own, MF_PIPE_*.)"""
        import ctypes
        import torch
        from . import _lib, pipeline
        dev = _torch_device_of(self)
        n = len(uncropped_frames)
        clip = pipeline.HostClip(uncropped_frames, n)
        H, W = clip.height, clip.width
        frames = [clip.array[i] for i in range(n)] if clip.array is not None else clip.frames       # (every frame validated by HostClip)
        left, top, right, bottom = (int(v) for v in crop_boundaries)
        out = np.empty((n, H, W, 3), dtype=np.uint8)
        fb = H * W * 3
        pin = (ctypes.c_void_p * n)(*[f.ctypes.data for f in frames])
        pout = (ctypes.c_void_p * n)(*[out.ctypes.data + i * fb for i in range(n)])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.mf_crop_resize_u8c3_host_frames(pin, pout, n, W, H, left, top, right, bottom, None))
        return list(out)

    def compute_scores(self, frame_width, frame_height, vertex_unstabilized_displacements_by_frame_index,
                       vertex_stabilized_displacements_by_frame_index, crop_boundaries):
        """(cropping_ratio, distortion_score, stability_score) in the order `stabilize` returns them (mfs.py:169).
        stability_score is the reference's own formula (mfs.py:1216-1259); the other two are computed from the
        mesh correspondences instead of tracked features -- see host.mesh_cropping_ratio_and_distortion."""
        ratio, distortion = host.mesh_cropping_ratio_and_distortion(
            frame_width, frame_height, self.mesh_row_count, self.mesh_col_count,
            vertex_unstabilized_displacements_by_frame_index, vertex_stabilized_displacements_by_frame_index,
            crop_boundaries)
        stab = np.asarray(vertex_stabilized_displacements_by_frame_index)
        return ratio, distortion, self._compute_stability_score(stab.shape[0], stab)

    def _compute_stability_score(self, num_frames, vertex_stabilized_displacements_by_frame_index):
        """mfs.py:1216-1259."""
        return host.stability_score(np.asarray(vertex_stabilized_displacements_by_frame_index))

    def _compute_stability_score_device(self, d_stab):
        """mfs.py:1216-1259 on device-resident paths (five direct DFT bins + Parseval; float64 rounding apart from
        `_compute_stability_score`, which stays the bit-for-bit restatement of the reference's np.fft formulation)."""
        from . import ops
        score, _ = ops.stability_score(d_stab)
        return float(score.item())

    def _get_vertex_x_y(self, frame_width, frame_height):
        """mfs.py:881-906."""
        return host.vertex_x_y(frame_width, frame_height, self.mesh_row_count, self.mesh_col_count)

    # ------------------------------------------------------------------------------------------
    # device-resident variants (torch tensors in HBM)
    # ------------------------------------------------------------------------------------------

    def _check_mesh_shape(self, disp, num_frames):
        want = (num_frames, self.mesh_row_count + 1, self.mesh_col_count + 1, 2)
        if tuple(disp.shape) != want:
            raise ValueError(f'vertex displacements must have shape {want}, got {tuple(disp.shape)}')

    def _jacobi_coefficients_device(self, num_frames, frame_width, frame_height, adaptive_weights_definition,
                                    homographies, device):
        """Band coefficients (host, O(F): mfs.py:713-841 restated in host.py) -> device.  The upload goes through a
        small ring of pinned buffers with a non-blocking copy: a pageable copy would wait for everything already
        queued on the stream, i.e. serialise this host work with the previous clip's kernels."""
        import torch
        taps, lam, inv_on = host.jacobi_band_coefficients(
            num_frames, frame_width, frame_height, adaptive_weights_definition,
            np.asarray(homographies, dtype=np.float64), self.temporal_smoothing_radius)
        nt = taps.size
        total = nt + 2 * num_frames
        ring = getattr(self, '_coef_ring', None)
        if ring is None or ring['size'] != total or ring['device'] != device:
            ring = {'size': total, 'device': device, 'next': 0,
                    'slots': [(torch.empty(total, dtype=torch.float64).pin_memory(), torch.cuda.Event()) for _ in range(4)]}
            self._coef_ring = ring
        staging, done = ring['slots'][ring['next']]
        ring['next'] = (ring['next'] + 1) % len(ring['slots'])
        done.synchronize()                                   # the copy that last used this slot (long finished)
        view = staging.numpy()
        view[:nt] = taps
        view[nt:nt + num_frames] = lam
        view[nt + num_frames:] = inv_on
        packed = torch.empty(total, dtype=torch.float64, device=device)
        packed.copy_(staging, non_blocking=True)
        done.record(torch.cuda.current_stream(device))
        return packed[:nt], packed[nt:nt + num_frames], packed[nt + num_frames:]

    def _stabilized_vertex_displacements_device(self, d_disp, frame_width, frame_height,
                                                adaptive_weights_definition, homographies):
        """d_disp: (F, R+1, C+1, 2) float64 device tensor -> same shape, stabilized."""
        from . import ops
        F = d_disp.shape[0]
        taps, lam, inv_on = self._jacobi_coefficients_device(F, frame_width, frame_height,
                                                             adaptive_weights_definition, homographies, d_disp.device)
        x = ops.jacobi(d_disp.reshape(F, -1), taps, lam, inv_on, self.temporal_smoothing_radius,
                       self.optimization_num_iterations)
        return x.view(d_disp.shape)

    # ---- the device-resident pipeline ----

    # How a resident clip is issued (measured on MI355X):
    #   0  (default) IN ORDER: cell table + plan, then the warp ALONE, on the caller's stream; only the Jacobi sweep goes to the prep
    #      stream, gated so that the NEXT clip's sweep runs beside THIS clip's table + plan (both leave most of the chip idle) and has
    #      ended when the warp starts.  Kernels that run beside the warp kernel cost it more than they take by themselves.
    #   k >= 1: the clip cut into k frame ranges, tables + crop scan + rectangle on the prep stream BESIDE the warp (warp(j) waits for
    #      table(j) only): the rectangle is known earliest; at the round-6 kernels a config-2 clip takes 1.88 ms this way against 1.19 (the
    #      side kernels are the YOUNGER wavefronts beside the warp -- cell table 273 us instead of 16, plan 468 instead of 70 -- and the
    #      stand-alone crop scan repeats the coordinate arithmetic; it was 2-8 % in round 4, before the in-order path lost a third of its time).
    resident_chunks = 0
    resident_rectangle = 'fused'     # 'early': rectangle from the table on the prep stream (a sharded run's all-reduce hides behind the warp)
    # What the NEXT clip's sweep (prep stream) waits for.  'table': this clip has reached its cell table -- a short sweep (config 2: 48 us)
    # then runs beside cell table + plan (~100 us, both leave most of the chip idle) and nothing runs beside the warp.  'plan': this clip's
    # plan has ended -- a LONG sweep (config 3: 0.65 ms of float64 FMAs on every SIMD) beside cell table + plan makes those take 0.7 + 0.4 ms
    # instead of 0.05 + 0.2; beside the first part of the warp it costs the warp what it takes itself, and the step is 7.5 % shorter
    # (3.73 -> 3.45 ms; config 2: 1.277 -> 1.269).  'auto': 'plan' from 4 GFLOP of sweep on.
    resident_gate = 'auto'
    resident_table_shapes = 4        # cell tables are kept for this many (W, H, mesh) geometries (two tables each, grow-only in the clip length)

    def _resident_state(self, dev):
        """Per-device plumbing of `stabilize_resident`: the PREP stream beside the caller's, a stream for the 4-byte status read-backs,
        two cell tables per clip geometry taking turns, the event after which a table is free again, and the gate of the next sweep."""
        import collections
        import torch
        st = getattr(self, '_resident', None)
        if st is None or st['device'] != dev:
            if st is not None:                               # another device: nothing of the old one's is kept
                self.finish()
            st = {'device': dev, 'prep': torch.cuda.Stream(device=dev), 'status': torch.cuda.Stream(device=dev),
                  'tables': collections.OrderedDict(), 'turn': 0, 'ends': [], 'serial': 0}
            self._resident = st
        return st

    @staticmethod
    def _settle(slot):
        """The degenerate-mesh verdict of the clip that last used this table slot: waits for its 4-byte status read-back (issued behind
        that clip's warp on a stream of its own -- long finished when the slot comes up again two clips later) and raises if the clip had
        cells without a homography.  The counter is cumulative per table; a slot remembers what it has seen."""
        pending, slot['pending'] = slot['pending'], None
        if pending is None:
            return
        pending['copied'].synchronize()
        total = int(slot['host'][0])
        bad, slot['seen'] = total - slot['seen'], total
        if bad and not pending['ignore']:
            raise DegenerateMeshError(bad, pending['serial'])

    def finish(self):
        """Waits for the verdict of every clip `stabilize_resident` has issued and not yet checked (it checks clip i when clip i + 2 is
        issued) and raises ValueError for the first one that had a degenerate mesh.  Later clips are unaffected either way."""
        st = getattr(self, '_resident', None)
        first = None
        if st is not None:
            for pair in st['tables'].values():
                for slot in pair:
                    try:
                        self._settle(slot)
                    except ValueError as e:
                        first = first or e
        if first is not None:
            raise first

    @property
    def resident_serial(self):
        """Serial number of the clip the last `stabilize_resident` call issued (what `DegenerateMeshError.clip_serial` names)."""
        st = getattr(self, '_resident', None)
        return 0 if st is None else st['serial']

    def _settle_upcoming(self, st, W, H):
        """The verdict of the clip whose table slot the NEXT clip of this geometry takes -- looked at before ANYTHING of the new clip is
        queued (its sweep included), so that a deferred DegenerateMeshError leaves no half-issued clip behind."""
        pair = st['tables'].get((W, H, self.mesh_row_count, self.mesh_col_count))
        if pair is not None:
            self._settle(pair[st['turn'] & 1])

    def _resident_slot(self, st, n, W, H):
        """The table slot of the next clip (two per geometry take turns), settled and sized for n frames."""
        import torch
        from . import ops
        dev = st['device']
        key = (W, H, self.mesh_row_count, self.mesh_col_count)
        pair = st['tables'].get(key)
        if pair is None:
            while len(st['tables']) >= max(1, self.resident_table_shapes):           # least recently used geometry out (its verdicts first)
                _, old = next(iter(st['tables'].items()))
                for slot in old:
                    self._settle(slot)
                torch.cuda.synchronize(dev)
                st['tables'].popitem(last=False)
            pair = st['tables'][key] = [{'table': ops.CellTable(n, W, H, self.mesh_row_count, self.mesh_col_count, dev), 'free': None,
                                         'pending': None, 'seen': 0, 'host': torch.zeros(1, dtype=torch.int32).pin_memory()} for _ in range(2)]
        st['tables'].move_to_end(key)
        slot = pair[st['turn'] & 1]
        self._settle(slot)                                   # (raises BEFORE anything of the new clip is issued or any state changes)
        if n > slot['table'].capacity:
            torch.cuda.synchronize(dev)                      # a longer clip than this slot has seen: its buffers are replaced (rare)
        slot['table'].resize(n)
        st['turn'] += 1
        return slot

    def _sweep_gate(self, d_disp):
        if self.resident_gate != 'auto':
            return self.resident_gate
        flops = float(self.optimization_num_iterations) * d_disp.shape[0] * (d_disp.numel() // d_disp.shape[0]) * (4 * self.temporal_smoothing_radius + 5)
        return 'plan' if flops >= 4e9 else 'table'

    def _resident_jacobi(self, d_disp, frame_width, frame_height, adaptive_weights_definition, homographies, inputs_ready=None):
        """Stage 1 of the resident pipeline, on the prep stream: mfs.py:632-710.  `inputs_ready`: a torch.cuda.Event after which
        d_disp (and the frames) are valid, or None = whatever is queued on the current stream right now (safe, but then this clip's
        sweep cannot start before the previous clip's warp has ended).  In-order mode gates the sweep on the previous clip having
        reached its cell table (= the warp before it has ended): it then runs beside that table + plan, not beside a warp."""
        import torch
        dev = d_disp.device
        st = self._resident_state(dev)
        prep = st['prep']
        if inputs_ready is None:
            inputs_ready = torch.cuda.Event()
            inputs_ready.record(torch.cuda.current_stream(dev))
        prep.wait_event(inputs_ready)
        # The gate is for sweeps of one wavefront per series (clips of up to 64 K frames: config 2's 47 us fit under cell table + plan,
        # config 3's 0.8 ms start there).  A longer clip spreads each series over 2-8 wavefronts with up to 40 KB of LDS per workgroup
        # (the replicated sweep of an N-GPU job): gated, it runs on into the warp and crowds it out of the CUs (rehearsal of a rank of 8:
        # 1.68 ms per step against 1.54 in order); left ungated, the queued sweeps of several clips fill the chip together (1.48).
        radius = self.temporal_smoothing_radius
        one_wave = d_disp.shape[0] <= 64 * (5 if radius <= 12 else 8 if radius <= 20 else 10)
        gate = self._sweep_gate(d_disp)
        st['gate'] = gate
        if self.resident_chunks <= 0 and one_wave and gate == 'plan' and st.get('planned') is not None:
            prep.wait_event(st['planned'])        # the previous clip's cell table + plan have ended: a LONG sweep runs beside its warp
        elif self.resident_chunks <= 0 and one_wave and len(st['ends']) >= 2:
            prep.wait_event(st['ends'][-2])       # the warp two clips back has ended = the previous clip has reached its cell table
        with torch.cuda.stream(prep):
            d_stab = self._stabilized_vertex_displacements_device(d_disp, frame_width, frame_height, adaptive_weights_definition, homographies)
            st['swept'] = torch.cuda.Event()
            st['swept'].record(prep)
        d_stab.record_stream(torch.cuda.current_stream(dev))      # allocated on the prep stream, handed to the caller's
        return d_stab

    def _resident_warp(self, d_frames, d_unstab, d_stab, out=None, chunks=None, warp_events=None, check='deferred'):
        """Stages 2-4, mfs.py:909-1108, for the d_stab `_resident_jacobi` just produced.  Returns (stabilized frames, bounds, table):
        bounds = the clip-level rectangle in a 16-byte tensor of this clip's own (the kernels fold it together there), table.crop per
        frame (valid until the table's next turn, two clips on).
        warp_events: two torch events recorded on the current stream in front of and behind the warp kernel(s) (bench.py's roofline)."""
        import torch
        from . import ops
        dev = d_frames.device
        st = self._resident_state(dev)
        main = torch.cuda.current_stream(dev)
        chunks = self.resident_chunks if chunks is None else chunks
        n, H, W, _ = d_frames.shape
        if n == 0:
            # an EMPTY shard (a clip of fewer frames than ranks, or the tail of an uneven split): nothing to warp; the rectangle's
            # neutral element (mfs.py:992-995), so that the rank still takes part in the all-reduce, and d_stab in stream order
            if st.get('swept') is not None:
                main.wait_event(st['swept'])
            bounds = torch.tensor([0, 0, W - 1, H - 1], dtype=torch.int32, device=dev)
            if self.resident_rectangle == 'early':
                st['scanned'] = torch.cuda.Event()
                st['scanned'].record(main)
            return (out if out is not None else d_frames.new_empty((0, H, W, 3))), bounds, None
        slot = self._resident_slot(st, n, W, H)
        table = slot['table']
        bounds = torch.empty(4, dtype=torch.int32, device=dev)           # this clip's own: never rewritten by a later one
        if chunks <= 0:
            if st.get('swept') is not None:
                main.wait_event(st['swept'])                      # the sweep that produced d_stab
            # (every marker on a stream costs the step ~5 us -- DESIGN.md section 5 -- so ONE event per clip, recorded behind its warp,
            # serves as the end of a timed interval, as "this table is free again", as the start of the status read-back and, two clips
            # on, as the gate of a sweep)
            # (the same launches as mf_warp_clip_u8c3 with chunks = 0, issued from here so that the warp kernel can be bracketed)
            ops.cell_table(d_unstab, d_stab, W, H, self.mesh_row_count, self.mesh_col_count, table=table, reset_status=False, bounds=bounds)
            if st.get('gate') == 'plan':
                st['planned'] = torch.cuda.Event()
                st['planned'].record(main)
            early = self.resident_rectangle == 'early'
            if early:                                             # rectangle from the table, on the prep stream, beside the start of the warp
                tabled = torch.cuda.Event()
                tabled.record(main)
                st['prep'].wait_event(tabled)
                with torch.cuda.stream(st['prep']):
                    ops.crop_scan(table, bounds=bounds)           # (the scan folds the clip-level rectangle together as well)
                    scanned = torch.cuda.Event()
                    scanned.record(st['prep'])
                st['scanned'] = scanned
            if warp_events:
                warp_events[0].record(main)
            out = ops.warp(d_frames, table, self.color_outside_image_area_bgr, out=out, bounds=bounds)
            end = warp_events[1] if warp_events else torch.cuda.Event()
            end.record(main)
            st['ends'].append(end)
            del st['ends'][:-2]
        else:
            if slot['free'] is not None:
                st['prep'].wait_event(slot['free'])               # the warps that last read this table (two clips ago) have ended
            if warp_events:
                warp_events[0].record(main)
            out, _ = ops.warp_clip(d_frames, d_unstab, d_stab, table, self.color_outside_image_area_bgr, out=out,
                                   chunks=chunks, prep_stream=st['prep'], bounds=bounds)
            end = warp_events[1] if warp_events else torch.cuda.Event()
            end.record(main)
        slot['free'] = end
        # the degenerate-cell counter of this clip, 4 bytes into pinned memory on a stream of their own behind the warp: no marker on
        # the caller's stream, nobody waits -- `_settle` looks at it when this slot comes up again (or `finish()` does); check='never'
        # only keeps the slot's running total in step
        ss = st['status']
        ss.wait_event(end)
        with torch.cuda.stream(ss):
            slot['host'].copy_(table.status, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(ss)
        st['serial'] += 1
        slot['pending'] = {'copied': copied, 'serial': st['serial'], 'ignore': check == 'never' or check is False}
        if check is True or check == 'now':
            self._settle(slot)
        return out, bounds, table

    def stabilize_resident(self, d_frames, d_disp, homographies, adaptive_weights_definition=ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL,
                           out=None, frame_range=None, inputs_ready=None, check=True, collective=False, warp_events=None,
                           jacobi_events=None):
        """mfs.py:150-158 for a clip whose frames (n, H, W, 3) uint8 and vertex displacements (F, R+1, C+1, 2) float64 are RESIDENT in
        HBM: Jacobi sweep -> cell tables -> warp + crop rectangle, nothing leaves the device, one call per clip, NO synchronisation:
        calls issued back to back pipeline by themselves (the sweep runs on this object's prep stream, the next clip's beside this
        clip's cell table + plan; see `resident_chunks` for the other arrangement).
        frame_range = (lo, hi): d_frames holds frames lo..hi-1 of the clip (a frame-range shard; the sweep still covers all F);
        collective=True then all-reduces the rectangle over the process group (16 bytes, `dist.allreduce_crop`).
        Returns (stabilized frames, clip-level crop bounds as a device int32 tensor {left, top, right, bottom} -- 16 bytes of this
        clip's own, never rewritten by a later call --, stabilized vertex displacements (F, R+1, C+1, 2)), all valid in current-stream
        order.
        A degenerate mesh (a cell without homography: the reference dies inside cv2) raises `DegenerateMeshError` (a ValueError) that
        carries the clip's serial number (`resident_serial` right after the call that issued it).  check=True (the default, like the
        reference: the error belongs to THIS call) waits for the clip's 4-byte verdict before returning -- one blocking device-to-host
        read per clip, so the calls no longer overlap.  A pipelined caller (bench.py's timed step) passes check='deferred': the verdict
        of clip i is then looked at when clip i + 2 is issued (its table slot comes up again; BEFORE anything of the new clip is queued,
        which is then not issued) or by `finish()`, whichever comes first -- such a caller MUST end with `finish()`.  check='never'
        skips it.  Clips after a degenerate one are unaffected.
        warp_events / jacobi_events: pairs of torch events recorded around the warp kernel (caller's stream) and the sweep stage (prep
        stream) -- bench.py's roofline brackets.
        One host thread per stabilizer object here: the calls of a pipeline are ordered by construction (table slots take turns, the
        deferred verdicts are kept per slot); threads that want their own pipelines take their own objects (the host-memory methods,
        `stabilize_clip` and the drop-in pair, may be called from several threads on one object: tests/test_gpu_stabilize_api.py)."""
        import torch
        from . import dist as mfdist
        self._check_definition(adaptive_weights_definition)
        if check is False:
            check = 'never'
        F = d_disp.shape[0]
        lo, hi = frame_range if frame_range is not None else (0, F)
        n, H, W, _ = d_frames.shape
        if hi - lo != n:
            raise ValueError(f'frame_range {lo, hi} does not match {n} frames')
        dev = d_frames.device
        st = self._resident_state(dev)
        if n > 0:
            self._settle_upcoming(st, W, H)              # (a deferred verdict is raised here: nothing of this clip has been queued yet)

        def jacobi_fn():
            if jacobi_events:
                jacobi_events[0].record(st['prep'])
            d_stab = self._resident_jacobi(d_disp, W, H, adaptive_weights_definition, homographies, inputs_ready)
            if jacobi_events:
                jacobi_events[1].record(st['prep'])
            return d_stab

        def warp_fn(lo_, hi_, d_stab):
            frames, bounds, _ = self._resident_warp(d_frames, d_disp[lo_:hi_], d_stab[lo_:hi_], out=out, warp_events=warp_events, check=check)
            return frames, bounds

        on_prep = self.resident_chunks > 0 or self.resident_rectangle == 'early'       # the rectangle is final on the prep stream, early

        def exchange_ctx():
            # the 16-byte all-reduce of a sharded clip goes to the prep stream too: beside the warp, off the critical path
            return torch.cuda.stream(st['prep'])

        frames, bounds, d_stab, _ = mfdist.stabilize_sharded(F, jacobi_fn, warp_fn, lambda b: b, frame_range=(lo, hi), collective=collective,
                                                             exchange_ctx=exchange_ctx if (on_prep and collective) else None)
        if on_prep:
            main = torch.cuda.current_stream(dev)
            if collective and mfdist.active():
                done = torch.cuda.Event()
                done.record(st['prep'])
                main.wait_event(done)                        # the exchanged rectangle, back in current-stream order
                bounds.record_stream(main)
            elif self.resident_chunks <= 0:
                main.wait_event(st['scanned'])               # (the chunked arrangement joins the streams inside mf_warp_clip_u8c3)
        return frames, bounds, d_stab

    def _stabilized_frames_device(self, d_frames, d_unstab, d_stab, out=None, table=None):
        """d_frames: (n, H, W, 3) uint8; d_unstab/d_stab: (n, R+1, C+1, 2) float64, all in HBM.
        Returns (stabilized frames (n, H, W, 3) uint8, per-frame crop values (n, 4) int32), in HBM."""
        from . import ops
        n, H, W, _ = d_frames.shape
        table = ops.cell_table(d_unstab, d_stab, W, H, self.mesh_row_count, self.mesh_col_count, table=table)
        out = ops.warp(d_frames, table, self.color_outside_image_area_bgr, out=out)
        table.check()
        return out, table.crop
