"""Multi-GPU orchestration: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by contiguous frame range:
  * Jacobi couples frames through a +-omega stencil for N sweeps, so the time axis is not shardable
    without a halo exchange per sweep; the whole (tiny) problem is REPLICATED on every GPU -- identical
    bits everywhere, no communication -- and each rank slices its frame range out of the result.
  * The warp is independent per frame: rank g warps frames [g*ceil(F/G), ...).
  * The only exchange the algorithm needs is the clip-level crop rectangle (mfs.py:1103-1106): one
    16-byte all-reduce, max over {left, top, -right, -bottom}.
  * `gather_frames` assembles the stabilized frames on one rank with a single gather over xGMI (padded to
    equal counts, as ncclGather requires).  It is optional: when the consumer is host memory, each rank
    draining its own shard through its own PCIe link is faster than funnelling everything through rank 0.

The compute steps are passed in as callables so that the same orchestration runs under gloo on CPU in
tests (with the oracle standing in for the kernels) and under RCCL on the GPU box.
"""
import os

import torch
import torch.distributed as dist

from . import host


def init_from_env(device_type=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, device)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if device_type is None:
        device_type = 'cuda' if torch.cuda.device_count() > 0 else 'cpu'
    # MESHFLOW_DIST_BACKEND=gloo lets several ranks share one GPU (functional testing of the N > 1 path on a
    # 1-GPU box); the default on GPUs is nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get('MESHFLOW_DIST_BACKEND', 'nccl' if device_type == 'cuda' else 'gloo')
    if device_type == 'cuda':
        index = local % torch.cuda.device_count() if backend != 'nccl' else local
        torch.cuda.set_device(index)
        device = torch.device('cuda', index)
    else:
        device = torch.device('cpu')
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        kwargs = {}
        if backend == 'nccl':
            kwargs['device_id'] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, device


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def active():
    """A process group exists (the collectives run, even over one rank)."""
    return dist.is_initialized()


def _through_host():
    """gloo moves tensors through host memory; device tensors are staged explicitly so that it works on every build."""
    return dist.is_initialized() and dist.get_backend() == 'gloo'


def all_reduce_max(t):
    """In-place MAX all-reduce of a small tensor on whatever backend is active.  (No process group: nothing to do.  A group of ONE rank
    still runs the collective -- the only way to exercise the RCCL path on a one-GPU box, tests/test_gpu_nccl_one_rank.py.)"""
    if not dist.is_initialized():
        return t
    if _through_host() and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


_SIGN = {}


def allreduce_crop(bounds):
    """bounds: int32 tensor {left, top, right, bottom} of this rank's frames -> clip-level bounds on every
    rank (max left, max top, min right, min bottom; mfs.py:1103-1106) with ONE 16-byte all-reduce."""
    if not dist.is_initialized():
        return bounds
    key = (bounds.device, bounds.dtype)
    sign = _SIGN.get(key)
    if sign is None:          # (made once: torch.tensor(..., device=cuda) is a blocking copy -- per step it would drain the stream)
        sign = _SIGN[key] = torch.tensor([1, 1, -1, -1], dtype=bounds.dtype, device=bounds.device)
    packed = bounds * sign
    all_reduce_max(packed)
    return packed * sign


def gather_frames(local_frames, num_frames, dst=0):
    """Gather the per-rank frame shards (rank g holds frames host.shard_range(F, G, g)) on rank `dst`.
    Returns the (F, H, W, 3) stack on dst and None elsewhere.  One collective; shards are padded to
    ceil(F/G) frames because gather needs equal counts."""
    G = world_size()
    if not dist.is_initialized():
        return local_frames
    rank = dist.get_rank()
    per = -(-num_frames // G)
    shape = (per,) + tuple(local_frames.shape[1:])
    if local_frames.shape[0] == per:
        send = local_frames.contiguous()
    else:
        send = torch.zeros(shape, dtype=local_frames.dtype, device=local_frames.device)
        send[:local_frames.shape[0]] = local_frames
    device = send.device
    if _through_host() and send.is_cuda:
        send = send.cpu()
    if rank == dst:
        full = torch.empty((G,) + shape, dtype=send.dtype, device=send.device)
        dist.gather(send, list(full.unbind(0)), dst=dst)
        return full.reshape((G * per,) + shape[1:])[:num_frames].to(device)
    dist.gather(send, None, dst=dst)
    return None


def stabilize_sharded(num_frames, jacobi_fn, warp_fn, crop_reduce_fn, gather=False, shard=None, collective=True, exchange_ctx=None,
                      frame_range=None):
    """Frame-range sharded pass of the hot path -- the function behind `MeshFlowStabilizer.stabilize_resident` (with the HIP
    operators; bench.py's timed step is that method) and what tests/test_dist_gloo.py drives under gloo (with the oracle standing in
    for the kernels).

    jacobi_fn() -> stabilized displacements of ALL frames (replicated on every rank)
    warp_fn(lo, hi, stab_all) -> (stabilized frames [hi-lo, H, W, 3], per-frame crop values [hi-lo, 4])
    crop_reduce_fn(per_frame_crop) -> int32 tensor {left, top, right, bottom} of this shard
    shard: (G, g) to take the frame range of rank g of G instead of this process's place in the process group (bench.py:
    independent clips = (1, 0); rehearsal of one rank of a larger job); collective=False skips the crop all-reduce
    (independent clips need none).  exchange_ctx: a callable returning a context manager under which the crop all-reduce is issued
    (the HIP pipeline passes its prep stream: the rectangle is final there before the warp ends, so the exchange runs beside the warp).
    Returns (local or gathered frames, clip-level crop bounds tensor, stab_all, (lo, hi))."""
    if frame_range is not None:                  # the caller holds frames lo..hi-1 (whatever the partition was)
        lo, hi = frame_range
    else:
        if shard is None:
            G = world_size()
            shard = (G, dist.get_rank() if G > 1 else 0)
        lo, hi = host.shard_range(num_frames, *shard)
    stab_all = jacobi_fn()
    frames, crop = warp_fn(lo, hi, stab_all)
    bounds = crop_reduce_fn(crop)
    if collective:
        if exchange_ctx is None:
            bounds = allreduce_crop(bounds)
        else:
            with exchange_ctx():
                bounds = allreduce_crop(bounds)
    if gather:
        frames = gather_frames(frames, num_frames)
    return frames, bounds, stab_all, (lo, hi)
