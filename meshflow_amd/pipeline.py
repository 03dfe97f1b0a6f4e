"""Chunked, overlapped host <-> device staging for clips that start and end in host memory.

SURVEY.md §8(f) row 4 (frame I/O staging), for the part that touches the GPU: the reference hands the hot path a
Python list of F NumPy frames and gets a list of F frames back (mfs.py:150-159, 213, 997-1100).  Moving 1.87 GB
each way (cfg2) through one blocking copy per direction costs 150x the kernels.  Here the clip is cut into
chunks of frames; a few host threads copy chunks in (each frame straight from its own array -- no `np.stack` of
the list), the warp of a chunk is issued as soon as its frames have landed, and other host threads copy finished
chunks out while later ones are still being uploaded and warped.  The three stages run on separate HIP streams
and are ordered by events, so PCIe carries both directions at once and the kernels hide behind the copies.

The Jacobi sweep needs only the (tiny) displacement tensor and runs before the first chunk arrives; the crop
rectangle is clip-level (mfs.py:1103-1106) and is reduced after the last chunk.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _as_frame(frame, height, width):
    a = np.ascontiguousarray(frame, dtype=np.uint8)
    if a.shape != (height, width, 3):
        raise ValueError(f'every frame must have shape ({height}, {width}, 3), got {a.shape}')
    return a


class HostClip:
    """A clip in host memory: either one (F, H, W, 3) uint8 array or a sequence of F (H, W, 3) arrays."""

    def __init__(self, frames, num_frames):
        if isinstance(frames, np.ndarray):
            if frames.ndim != 4 or frames.shape[0] != num_frames or frames.shape[3] != 3:
                raise ValueError('frames must be num_frames arrays of shape (H, W, 3)')
            self.array = np.ascontiguousarray(frames, dtype=np.uint8)
            self.frames = None
            self.height, self.width = self.array.shape[1:3]
        else:
            if len(frames) != num_frames or num_frames == 0:
                raise ValueError('frames must be num_frames arrays of shape (H, W, 3)')
            first = np.asarray(frames[0])
            if first.ndim != 3 or first.shape[2] != 3:
                raise ValueError('frames must be num_frames arrays of shape (H, W, 3)')
            self.array = None
            self.frames = frames
            self.height, self.width = first.shape[:2]
        self.num_frames = num_frames

    def upload(self, d_frames, i0, i1):
        """Blocking copy of frames i0..i1-1 into d_frames[i0:i1] on the calling thread's current stream."""
        import torch
        if self.array is not None:
            d_frames[i0:i1].copy_(torch.from_numpy(self.array[i0:i1]))
        else:
            for i in range(i0, i1):
                d_frames[i].copy_(torch.from_numpy(_as_frame(self.frames[i], self.height, self.width)))


def chunk_ranges(num_frames, chunk_frames):
    chunk_frames = max(1, int(chunk_frames))
    return [(i, min(i + chunk_frames, num_frames)) for i in range(0, num_frames, chunk_frames)]


class ChunkedTransfer:
    """Thread pools + streams for the two copy directions.  `upload_all` starts every upload and returns one
    (future -> event) per chunk, in order; `download` queues the copy of a device chunk into a host array once
    `after` (an event on the compute stream) has fired."""

    def __init__(self, device, in_threads=3, out_threads=3):
        import torch
        self.device = device
        self.in_streams = [torch.cuda.Stream(device=device) for _ in range(max(1, in_threads))]
        self.out_streams = [torch.cuda.Stream(device=device) for _ in range(max(1, out_threads))]
        self.in_pool = ThreadPoolExecutor(max_workers=len(self.in_streams), thread_name_prefix='mf-h2d')
        self.out_pool = ThreadPoolExecutor(max_workers=len(self.out_streams), thread_name_prefix='mf-d2h')
        self.pending = []

    def upload(self, clip, d_frames, i0, i1, after, k):
        """Queues the copy of clip frames i0..i1-1 into d_frames[i0:i1] on copy stream k (mod the pool); returns a
        future whose result is the event that marks its end.  `after`: event on the stream that allocated d_frames
        (the copy streams must not run ahead of it)."""
        import torch

        def task():
            stream = self.in_streams[k % len(self.in_streams)]
            with torch.cuda.device(self.device), torch.cuda.stream(stream):
                stream.wait_event(after)
                clip.upload(d_frames, i0, i1)
                ev = torch.cuda.Event()
                ev.record(stream)
            return ev

        return self.in_pool.submit(task)

    def upload_all(self, clip, d_frames, ranges, after):
        return [self.upload(clip, d_frames, i0, i1, after, k) for k, (i0, i1) in enumerate(ranges)]

    def download(self, d_src, host_dst, after, k):
        import torch

        def task():
            stream = self.out_streams[k % len(self.out_streams)]
            with torch.cuda.device(self.device), torch.cuda.stream(stream):
                stream.wait_event(after)
                torch.from_numpy(host_dst).copy_(d_src)          # blocking on this thread, this stream

        self.pending.append(self.out_pool.submit(task))

    def finish(self):
        """Wait for every queued download; re-raises the first failure."""
        pending, self.pending = self.pending, []
        for f in pending:
            f.result()

    def close(self):
        self.in_pool.shutdown(wait=True)
        self.out_pool.shutdown(wait=True)
