"""The stages of `stabilize()` that the reference delegates to OpenCV, delegated to OpenCV here as well.

Outside the MI355X path (video codec, feature tracker and feature-based metrics are out of scope),
kept only so that `MeshFlowStabilizer.stabilize(input_path, output_path)` is a working drop-in wherever `cv2` is
installed.  Behaviour follows /root/reference/meshflowstabilizer.py (`mfs.py`):

  read_video            mfs.py:172-233    cv2.VideoCapture -> list of BGR frames, fps, fourcc; IOError on a short file
  track_pair            mfs.py:455-629    FAST corners per sub-frame, pyramidal LK, RANSAC outlier rejection per
                                          sub-frame, one least-squares homography over the survivors
  cropping_and_distortion  mfs.py:1160-1212  the two feature-based scores
  write_video / show_loop  mfs.py:1262-1322

`cv2` is imported lazily and passed in, so importing this module never needs it.  Frame pairs are independent and
OpenCV releases the GIL, so pairs are tracked on a small thread pool (the reference's "TODO parallelize").
"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def require_cv2():
    try:
        import cv2
    except ImportError as e:
        raise ImportError(
            'MeshFlowStabilizer.stabilize(path, path) needs OpenCV (cv2) for video decode/encode and the '
            'FAST/LK/RANSAC tracker, which are outside the MI355X path. Use stabilize_clip(frames, '
            'vertex_unstabilized_displacements, homographies) or _get_unstabilized_vertex_displacements_from_features '
            'with in-memory inputs instead.') from e
    return cv2


def read_video(cv2, input_path):
    """(frames, num_frames, frames_per_second, fourcc) -- mfs.py:193-213."""
    video = cv2.VideoCapture(input_path)
    try:
        num_frames = int(video.get(cv2.CAP_PROP_FRAME_COUNT))
        fps = video.get(cv2.CAP_PROP_FPS)
        codec = int(video.get(cv2.CAP_PROP_FOURCC))
        frames = []
        for index in range(num_frames):
            ok, pixels = video.read()
            if not ok:
                raise IOError(f'Video at <{input_path}> did not have frame {index} of {num_frames} (indexed from 0).')
            frames.append(pixels)
    finally:
        video.release()
    return frames, num_frames, fps, codec


def write_video(cv2, output_path, frames_per_second, codec, frames):
    """mfs.py:1309-1322."""
    height, width = frames[0].shape[:2]
    video = cv2.VideoWriter(output_path, codec, frames_per_second, (width, height))
    try:
        for frame in frames:
            video.write(frame)
    finally:
        video.release()


def show_loop(cv2, frames_per_second, unstabilized_frames, cropped_frames):
    """mfs.py:1282-1287: both clips stacked, looping until Q is pressed."""
    delay = int(1000 / frames_per_second)
    while True:
        for before, after in zip(unstabilized_frames, cropped_frames):
            cv2.imshow('unstabilized and stabilized video', np.vstack((before, after)))
            if cv2.waitKey(delay) & 0xFF == ord('q'):
                return


class Tracker:
    """FAST + LK + per-sub-frame RANSAC, with the stabilizer's sub-frame grid and minimum feature count."""

    def __init__(self, cv2, subframe_rows, subframe_cols, min_features):
        self.cv2 = cv2
        self.subframe_rows = subframe_rows
        self.subframe_cols = subframe_cols
        self.min_features = min_features

    def _subframe_features(self, early, late, offset):
        """mfs.py:531-629 for one sub-frame; (None, None) when too few features survive a stage."""
        cv2 = self.cv2
        keypoints = cv2.FastFeatureDetector_create().detect(early)      # a detector per call: detect() is not re-entrant
        if len(keypoints) < self.min_features:
            return None, None
        points = np.float32(cv2.KeyPoint_convert(keypoints)[:, np.newaxis, :])
        moved, found, _ = cv2.calcOpticalFlowPyrLK(early, late, points, None)
        keep = found.flatten().astype(bool)
        points, moved = points[keep], moved[keep]
        if len(points) < self.min_features:
            return None, None
        _, inliers = cv2.findHomography(points, moved, method=cv2.RANSAC)
        keep = inliers.flatten().astype(bool)
        # adding the (int, int) offset promotes the float32 coordinates to float64, as in mfs.py:578
        return points[keep] + offset, moved[keep] + offset

    def track_pair(self, early_frame, late_frame):
        """(early_features, late_features, homography) of one frame pair, or (None, None, None) -- mfs.py:492-528."""
        height, width = early_frame.shape[:2]
        sub_w = math.ceil(width / self.subframe_cols)
        sub_h = math.ceil(height / self.subframe_rows)
        early_parts, late_parts = [], []
        for left in range(0, width, sub_w):
            for top in range(0, height, sub_h):
                e, l = self._subframe_features(early_frame[top:top + sub_h, left:left + sub_w],
                                               late_frame[top:top + sub_h, left:left + sub_w], [left, top])
                if e is not None:
                    early_parts.append(e)
                    late_parts.append(l)
        if not early_parts:
            return None, None, None
        early, late = np.concatenate(early_parts), np.concatenate(late_parts)
        if len(early) < self.min_features:
            return None, None, None
        homography, _ = self.cv2.findHomography(early, late)
        return early, late, homography

    def track_pairs(self, first_frames, second_frames, workers=8):
        """track_pair over many independent pairs, in order."""
        with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
            return list(pool.map(self.track_pair, first_frames, second_frames))


def cropping_and_distortion(tracker, unstabilized_frames, cropped_frames, workers=8):
    """(cropping_ratio, distortion_score) -- mfs.py:1187-1212: per frame, the homography from the unstabilized to
    the cropped frame; ratio = 1 / (h00 * h11), distortion = ratio of the two largest eigenvalue magnitudes of its
    affine part; float32 mean of the ratios, float32 MINIMUM of the distortions (the reference's choice)."""
    tracked = tracker.track_pairs(unstabilized_frames, cropped_frames, workers)
    return cropping_and_distortion_from_homographies([h for _, _, h in tracked])


def cropping_and_distortion_from_homographies(homographies):
    """The arithmetic of mfs.py:1196-1212 on the per-frame (unstabilized -> cropped) homographies."""
    ratios = np.empty(len(homographies), dtype=np.float32)
    distortions = np.empty(len(homographies), dtype=np.float32)
    for i, h in enumerate(homographies):
        ratios[i] = 1 / (h[0][0] * h[1][1])        # h is None when tracking failed: TypeError, like the reference
        affine = np.copy(h)
        affine[2] = [0, 0, 1]
        magnitudes = np.sort(np.abs(np.linalg.eigvals(affine)))
        distortions[i] = magnitudes[-2] / magnitudes[-1]
    return np.mean(ratios), np.min(distortions)
