"""Host-side glue around OpenCV (`meshflow_amd/frontend_cv2.py`) against tests/fake_cv2.py -- no GPU, no OpenCV."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_cv2  # noqa: E402

from meshflow_amd import frontend_cv2  # noqa: E402


@pytest.fixture()
def cv2():
    fake_cv2.VIDEOS.clear(); fake_cv2.WRITTEN.clear(); fake_cv2.MOTION.clear()
    return fake_cv2.module()


def test_track_pair_offsets_and_dtype(cv2):
    early = np.zeros((60, 80, 3), np.uint8)
    late = np.zeros((60, 80, 3), np.uint8)
    fake_cv2.MOTION[id(late)] = (2.0, -1.0)
    tr = frontend_cv2.Tracker(cv2, 2, 2, 4)
    e, l, h = tr.track_pair(early, late)
    assert e.dtype == np.float64 and l.dtype == np.float64 and e.shape == l.shape and e.shape[1:] == (1, 2)
    assert e[:, 0, 0].max() > 40 and e[:, 0, 1].max() > 30            # sub-frame offsets were added back (mfs.py:578)
    assert abs(h[0, 2] - 2.0) < 0.1 and abs(h[1, 2] + 1.0) < 0.1 and abs(h[0, 0] - 1.0) < 1e-2       # the registered shift
    # too few features anywhere -> (None, None, None), mfs.py:521-522
    assert frontend_cv2.Tracker(cv2, 2, 2, 10 ** 6).track_pair(early, late) == (None, None, None)


def test_track_pairs_keeps_order(cv2):
    frames = [np.zeros((40, 40, 3), np.uint8) for _ in range(6)]
    for t, f in enumerate(frames):
        fake_cv2.MOTION[id(f)] = (float(t), 0.0)
    tr = frontend_cv2.Tracker(cv2, 1, 1, 4)
    out = tr.track_pairs(frames[:-1], frames[1:], workers=3)
    assert [round(float(h[0, 2])) for _, _, h in out] == [1, 2, 3, 4, 5]


def test_read_write_video(cv2):
    frames = [np.full((8, 10, 3), i, np.uint8) for i in range(4)]
    fake_cv2.VIDEOS['a'] = dict(frames=frames, fps=25.0, fourcc=7, claimed=None)
    got, n, fps, codec = frontend_cv2.read_video(cv2, 'a')
    assert n == 4 and fps == 25.0 and codec == 7 and all(a is b for a, b in zip(got, frames))
    fake_cv2.VIDEOS['b'] = dict(frames=frames, fps=25.0, fourcc=7, claimed=6)
    with pytest.raises(IOError, match='did not have frame 4 of 6'):
        frontend_cv2.read_video(cv2, 'b')
    frontend_cv2.write_video(cv2, 'out', 25.0, 7, frames)
    rec = fake_cv2.WRITTEN['out']
    assert rec['size'] == (10, 8) and len(rec['frames']) == 4


def test_cropping_and_distortion_formulas(cv2):
    class T:
        def track_pairs(self, a, b, workers=8):
            return [(None, None, np.array([[1.25, 0.0, 3.0], [0.0, 1.6, -2.0], [0.0, 0.0, 1.0]])),
                    (None, None, np.array([[1.0, 0.0, 0.0], [0.0, 2.0, 0.0], [1e-4, 0.0, 1.0]]))]
    ratio, distortion = frontend_cv2.cropping_and_distortion(T(), [0, 1], [0, 1])
    assert isinstance(ratio, np.float32) and isinstance(distortion, np.float32)
    assert ratio == np.mean(np.array([1 / (1.25 * 1.6), 1 / 2.0], np.float32))
    # eigenvalue magnitudes {1, 1.25, 1.6} -> 1.25/1.6 ; {1, 1, 2} -> 1/2 ; the reference takes the MINIMUM (mfs.py:1212)
    assert distortion == np.float32(0.5)


def test_stabilize_says_what_is_missing_without_cv2(monkeypatch):
    monkeypatch.setitem(sys.modules, 'cv2', None)
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    with pytest.raises(ImportError, match='needs OpenCV'):
        MeshFlowStabilizer().stabilize('in.m4v', 'out.m4v')
