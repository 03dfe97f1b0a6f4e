"""`stabilize(input_path, output_path)` with the frame I/O overlapped with everything around it.

The reference decodes the whole video, then tracks, smooths, warps, crops, scores and finally encodes, strictly one
stage after the other (mfs.py:148-167).  The stages' data dependencies allow three overlaps, and this module takes them:

  while decoding (cv2.VideoCapture.read, mfs.py:193-213)
      * frame pair (t-1, t) goes to the tracker pool as soon as frame t is decoded (mfs.py:268-282 needs every pair, but the
        pairs are independent);
      * every complete chunk of frames is uploaded to the GPU on a copy stream (the warp needs them later; PCIe is idle now);
  then the dense path on the device, frames already resident: vertex displacements from the features, Jacobi (needs ALL
  frames' displacements, so it cannot start earlier), cell tables + warp per chunk, clip-level crop bounds, crop + resize;
  while the cropped frames come back chunk by chunk
      * each chunk is handed to the encoder (cv2.VideoWriter.write, mfs.py:1309-1322) as soon as it has landed, in order,
        while later chunks are still on the link;
      * the two feature-based scores (mfs.py:1160-1212) track (unstabilized, cropped) pairs of the chunks that are back.

OpenCV stays OpenCV (decode, tracker, encode: `frontend_cv2.py`); nothing here restates it.  Used by
`MeshFlowStabilizer.stabilize` when no boundary method is overridden; exercised with `tests/fake_cv2.py`.
"""
import queue
from concurrent.futures import ThreadPoolExecutor

import numpy as np


# Decoder-paced staging: frames arrive ONE AT A TIME from cv2.VideoCapture and leave one chunk at a time towards cv2.VideoWriter, so
# this module feeds the device as chunks complete (the C ring of csrc/hostpipe.hip serves calls that hold the whole clip).

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



def stabilize_streamed(stab, cv2, input_path, output_path, adaptive_weights_definition, chunk_frames=16, workers=8,
                       io_threads=3):
    """Returns (cropping_ratio, distortion_score, stability_score) and writes the stabilized video -- mfs.py:102-169."""
    import torch
    from . import frontend_cv2, host, ops, pipeline
    dev = stab._torch_device()
    tracker = stab._tracker()
    io = ChunkedTransfer(dev, io_threads, io_threads)
    pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix='mf-track')
    video = cv2.VideoCapture(input_path)
    try:
        # ---- decode || track || upload ------------------------------------------------------------------------------
        num_frames = int(video.get(cv2.CAP_PROP_FRAME_COUNT))
        frames_per_second = video.get(cv2.CAP_PROP_FPS)
        codec = int(video.get(cv2.CAP_PROP_FOURCC))
        ranges = chunk_ranges(num_frames, chunk_frames)
        frames, pair_futures, uploads = [], [], []
        d_frames = allocated = None
        next_chunk = 0
        for index in range(num_frames):
            ok, pixels = video.read()
            if not ok:
                raise IOError(f'Video at <{input_path}> did not have frame {index} of {num_frames} (indexed from 0).')
            frames.append(pixels)
            if index > 0:
                pair_futures.append(pool.submit(tracker.track_pair, frames[index - 1], pixels))
            if d_frames is None:
                H, W = pixels.shape[:2]
                d_frames = torch.empty((num_frames, H, W, 3), dtype=torch.uint8, device=dev)
                allocated = torch.cuda.Event()
                allocated.record(torch.cuda.current_stream(dev))
            if next_chunk < len(ranges) and index + 1 == ranges[next_chunk][1]:
                i0, i1 = ranges[next_chunk]
                clip = pipeline.HostClip(frames[i0:i1], i1 - i0)
                if (clip.height, clip.width) != (H, W):
                    raise ValueError(f'every frame must have shape ({H}, {W}, 3)')
                uploads.append(io.upload(clip, d_frames[i0:i1], 0, i1 - i0, allocated, next_chunk))
                next_chunk += 1
        video.release()
        video = None
        if num_frames < 1:
            raise IOError(f'Video at <{input_path}> has no frames.')

        tracked = [f.result() for f in pair_futures]
        homographies = np.empty((num_frames, 3, 3))
        homographies[-1] = np.identity(3)                                                   # mfs.py:274
        for t, (_, _, h) in enumerate(tracked):
            if h is None:
                raise ValueError(f'fewer than {stab.homography_min_number_corresponding_features} features could be '
                                 f'tracked from frame {t} to frame {t + 1}')
            homographies[t] = h

        # ---- the dense path, device-resident ------------------------------------------------------------------------
        R, C = stab.mesh_row_count, stab.mesh_col_count
        early, late, offsets, kmax = host.pack_features([(e, l) for e, l, _ in tracked])
        d_unstab, _, status = ops.vertex_motion(
            torch.from_numpy(early).to(dev), torch.from_numpy(late).to(dev), torch.from_numpy(offsets).to(dev),
            torch.from_numpy(np.ascontiguousarray(homographies[:num_frames - 1]).reshape(-1, 3, 3)).to(dev), kmax, W, H, R, C,
            stab.feature_ellipse_row_count, stab.feature_ellipse_col_count)
        ops.vertex_motion_check(status)
        d_stab = stab._stabilized_vertex_displacements_device(d_unstab, W, H, adaptive_weights_definition, homographies)
        d_out = torch.empty_like(d_frames)
        d_crop = torch.empty((num_frames, 4), dtype=torch.int32, device=dev)
        compute = torch.cuda.current_stream(dev)
        tables = {}
        for k, (i0, i1) in enumerate(ranges):
            compute.wait_event(uploads[k].result())
            table = ops.cell_table(d_unstab[i0:i1], d_stab[i0:i1], W, H, R, C, table=tables.get(i1 - i0), reset_status=False)
            tables[i1 - i0] = table
            ops.warp(d_frames[i0:i1], table, stab.color_outside_image_area_bgr, out=d_out[i0:i1])
            d_crop[i0:i1].copy_(table.crop)
        bounds = tuple(np.int64(v) for v in ops.crop_reduce(d_crop, W, H).cpu().numpy())
        for table in tables.values():
            table.check()
        d_cropped = ops.crop_resize(d_out, bounds, out=d_frames)         # the unstabilized stack on the device is no longer needed
        stability_score = stab._compute_stability_score(num_frames, d_stab.cpu().numpy())

        # ---- download || encode || feature-based scores -------------------------------------------------------------
        cropped = np.empty((num_frames, H, W, 3), dtype=np.uint8)
        done = torch.cuda.Event()
        done.record(compute)
        landed = queue.Queue()

        def fetch(k, i0, i1):
            io.download(d_cropped[i0:i1], cropped[i0:i1], done, k)
            fut = io.pending[-1]
            fut.add_done_callback(lambda f, k=k: landed.put((k, f.exception())))

        for k, (i0, i1) in enumerate(ranges):
            fetch(k, i0, i1)
        writer = cv2.VideoWriter(output_path, codec, frames_per_second, (W, H))
        score_futures = [None] * num_frames
        try:
            arrived, next_write = {}, 0
            while next_write < len(ranges):
                k, err = landed.get()
                if err is not None:
                    raise err
                arrived[k] = True
                i0, i1 = ranges[k]
                for i in range(i0, i1):                              # mfs.py:1195: tracker on (unstabilized, cropped) pairs
                    score_futures[i] = pool.submit(tracker.track_pair, frames[i], cropped[i])
                while next_write < len(ranges) and arrived.get(next_write):
                    j0, j1 = ranges[next_write]
                    for i in range(j0, j1):
                        writer.write(cropped[i])
                    next_write += 1
        finally:
            writer.release()
        io.finish()
        cropping_ratio, distortion_score = frontend_cv2.cropping_and_distortion_from_homographies(
            [f.result()[2] for f in score_futures])
        if stab.visualize:
            frontend_cv2.show_loop(cv2, frames_per_second, frames, list(cropped))
        return cropping_ratio, distortion_score, stability_score
    finally:
        if video is not None:
            video.release()
        pool.shutdown(wait=True)
        io.close()
