"""Known-answer tests for the oracle's restatement of cv2.resize INTER_LINEAR on uint8 (mfs.py:1150-1155).
Parity unpinned (no cv2 here): these are properties any correct restatement of resize.cpp must have."""
import numpy as np

from meshflow_amd import synthetic
from oracle import meshflow_oracle as mo


def test_same_size_is_identity():
    fr = synthetic.frames_numpy(1, 40, 60, seed=1, kind='noise')[0]
    np.testing.assert_array_equal(mo.resize_linear_u8(fr, 60, 40), fr)


def test_constant_image_stays_constant():
    for val in (0, 1, 77, 254, 255):
        c = np.full((10, 13, 3), val, np.uint8)
        assert np.unique(mo.resize_linear_u8(c, 60, 40)).tolist() == [val]


def test_two_times_upscale_of_a_ramp():
    # pixel centres: dst x samples src (x + 0.5)/2 - 0.5 = -0.25, 0.25, 0.75, ...; ends clamp to the border pixel
    r = (np.arange(8, dtype=np.uint8) * 10)[None, :, None].repeat(4, 0).repeat(3, 2)
    got = mo.resize_linear_u8(r, 16, 4)[0, :, 0]
    np.testing.assert_array_equal(got, [0, 3, 8, 13, 18, 23, 28, 33, 38, 43, 48, 53, 58, 63, 68, 70])


def test_tables_follow_the_half_pixel_convention():
    s, f = mo.resize_linear_tables(100, 120)
    scale = 1.0 / (120.0 / 100.0)
    exp = (np.arange(120) + 0.5) * scale - 0.5
    np.testing.assert_array_equal(s, np.floor(exp.astype(np.float32)))
    assert np.abs((s + f) - exp).max() < 1e-5
    assert s[0] == -1 and s[-1] == 99          # both ends leave the source and are clamped by the caller


def test_crop_frames_matches_manual_crop_then_resize():
    frames = list(synthetic.frames_numpy(2, 48, 64, seed=3, kind='noise'))
    out = mo.crop_frames(frames, (5, 3, 60, 44))
    assert len(out) == 2 and out[0].shape == (48, 64, 3)
    np.testing.assert_array_equal(out[1], mo.resize_linear_u8(frames[1][3:45, 5:61], 64, 48))
    # separable: resizing rows then columns with exact intermediate equals the two-pass result on a
    # horizontally constant image
    col = frames[0][:, :1].repeat(64, 1)
    a = mo.resize_linear_u8(col[3:45, 5:61], 64, 48)
    assert (a == a[:, :1]).all()
