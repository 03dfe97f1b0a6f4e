"""A clip in host memory, as the reference hands it to the hot path: a Python list of F NumPy frames (mfs.py:150-159, 213,
997-1100) or one (F, H, W, 3) array.  The staging itself -- chunked, overlapped PCIe copies around the kernels -- lives below
Python, in csrc/hostpipe.hip (ONE implementation: `mf_warp_u8c3_host_frames`, `mf_warp_crop_u8c3_host_frames`,
`mf_crop_resize_u8c3_host_frames`); this module only validates the frames and hands out their addresses."""
import numpy as np


def _need_u8(a):
    """cv2.remap runs its 8-bit fixed-point path on uint8 frames only (mfs.py:1063-1069: what a VideoCapture delivers); float or 16-bit
    frames would take other OpenCV code with other arithmetic -- refused instead of silently cast."""
    if a.dtype != np.uint8:
        raise TypeError(f'frames must be uint8 (got {a.dtype}): the warp reproduces cv2.remap\'s 8-bit fixed-point interpolation')


def _as_frame(frame, height, width):
    a = np.asarray(frame)
    _need_u8(a)
    a = np.ascontiguousarray(a)
    if a.shape != (height, width, 3):
        raise ValueError(f'every frame must have shape ({height}, {width}, 3), got {a.shape}')
    return a


class HostClip:
    """A clip in host memory: either one (F, H, W, 3) uint8 array or a sequence of F (H, W, 3) arrays."""

    def __init__(self, frames, num_frames):
        if isinstance(frames, np.ndarray):
            if frames.ndim != 4 or frames.shape[0] != num_frames or frames.shape[3] != 3:
                raise ValueError('frames must be num_frames arrays of shape (H, W, 3)')
            _need_u8(frames)
            self.array = np.ascontiguousarray(frames)
            self.frames = None
            self.height, self.width = self.array.shape[1:3]
        else:
            if len(frames) != num_frames or num_frames == 0:
                raise ValueError('frames must be num_frames arrays of shape (H, W, 3)')
            first = np.asarray(frames[0])
            if first.ndim != 3 or first.shape[2] != 3:
                raise ValueError('frames must be num_frames arrays of shape (H, W, 3)')
            self.height, self.width = first.shape[:2]
            # EVERY frame is checked here, before anything is issued (dtype uint8, the first frame's shape; C-contiguous -- a frame that
            # is not gets a contiguous copy, the others are taken as they are): a bad frame k fails in the constructor, not mid-clip
            self.array = None
            self.frames = [_as_frame(f, self.height, self.width) for f in frames]
        self.num_frames = num_frames

    def upload(self, d_frames, i0, i1):
        """Blocking copy of frames i0..i1-1 into d_frames[i0:i1] on the calling thread's current stream (streaming.py's decode-side
        staging; the clip methods of the stabilizer go through csrc/hostpipe.hip instead)."""
        import torch
        if self.array is not None:
            d_frames[i0:i1].copy_(torch.from_numpy(self.array[i0:i1]))
        else:
            for i in range(i0, i1):
                d_frames[i].copy_(torch.from_numpy(self.frames[i]))
