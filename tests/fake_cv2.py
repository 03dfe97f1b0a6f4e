"""A tiny stand-in for the handful of OpenCV entry points `meshflow_amd/frontend_cv2.py` calls, so that the
orchestration of `MeshFlowStabilizer.stabilize(input_path, output_path)` can be exercised where OpenCV is not
installed.  It is NOT an OpenCV restatement: the "detector" returns a fixed lattice of points, the "optical flow"
moves them by a shift registered for the late frame, the "homography" is a plain least-squares DLT."""
import threading
import time
import types

import numpy as np

VIDEOS = {}      # path -> dict(frames=[...], fps=float, fourcc=int, claimed=int or None)
WRITTEN = {}     # path -> dict(frames=[...], fps, fourcc, size)
MOTION = {}      # id(frame array) -> (dx, dy) applied by calcOpticalFlowPyrLK when that frame is the late one
EVENTS = []      # ('read', i) / ('write', i) in the order they happened (tests append their own kinds through `log`)
READ_DELAY = [0.0]   # seconds a decode of one frame takes
_LOCK = threading.Lock()


def log(kind, index):
    with _LOCK:
        EVENTS.append((kind, index))


class _Capture:
    def __init__(self, path):
        self.video = VIDEOS[path]
        self.pos = 0

    def get(self, prop):
        v = self.video
        return {1: float(v.get('claimed') or len(v['frames'])), 2: v['fps'], 3: float(v['fourcc'])}[prop]

    def read(self):
        if self.pos >= len(self.video['frames']):
            return False, None
        if READ_DELAY[0]:
            time.sleep(READ_DELAY[0])
        self.pos += 1
        log('read', self.pos - 1)
        return True, self.video['frames'][self.pos - 1]

    def release(self):
        pass


class _Writer:
    def __init__(self, path, fourcc, fps, size):
        self.rec = WRITTEN[path] = dict(frames=[], fps=fps, fourcc=fourcc, size=size)

    def write(self, frame):
        log('write', len(self.rec['frames']))
        self.rec['frames'].append(np.array(frame, copy=True))

    def release(self):
        pass


class _Detector:
    def detect(self, image):
        h, w = image.shape[:2]
        return [types.SimpleNamespace(pt=(float(x), float(y))) for y in range(3, h - 3, 7) for x in range(4, w - 4, 9)]


def _owner(a):
    while getattr(a, 'base', None) is not None:
        a = a.base
    return a


def _flow(early, late, points, _next):
    dx, dy = MOTION.get(id(_owner(late)), (0.0, 0.0))
    x, y = points[:, 0, 0], points[:, 0, 1]
    moved = points + np.stack([dx + 0.002 * y, dy - 0.001 * x], axis=-1)[:, None, :].astype(np.float32)
    return moved.astype(np.float32), np.ones((len(points), 1), np.uint8), None


def _find_homography(src, dst, method=0):
    s = np.asarray(src, np.float64).reshape(-1, 2)
    d = np.asarray(dst, np.float64).reshape(-1, 2)
    rows = []
    for (x, y), (u, v) in zip(s, d):
        rows.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
        rows.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
    _, _, vt = np.linalg.svd(np.asarray(rows))
    h = vt[-1].reshape(3, 3)
    return h / h[2, 2], np.ones((len(s), 1), np.uint8)


def module():
    m = types.ModuleType('cv2')
    m.CAP_PROP_FRAME_COUNT, m.CAP_PROP_FPS, m.CAP_PROP_FOURCC, m.RANSAC = 1, 2, 3, 8
    m.VideoCapture, m.VideoWriter = _Capture, _Writer
    m.FastFeatureDetector_create = _Detector
    m.KeyPoint_convert = lambda kps: np.array([k.pt for k in kps], np.float32)
    m.calcOpticalFlowPyrLK = _flow
    m.findHomography = _find_homography
    m.__fake__ = True
    return m
