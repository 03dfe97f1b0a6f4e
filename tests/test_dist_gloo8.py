"""The multi-GPU orchestration (meshflow_amd/dist.py) at the size of the real node: EIGHT gloo ranks on CPU.

What breaks first on an 8-GPU node is not the kernels but the partition: 300 frames over 8 ranks are 7 shards of 38 and one of 34,
12 frames leave two ranks with NOTHING (their crop contribution must be the neutral element, their gather slot pure padding),
2400 frames = BASELINE config 4's 8 x 300.  The oracle stands in for the HIP kernels (they cannot run here); everything else --
`host.shard_range`, `dist.stabilize_sharded`, the 16-byte crop all-reduce, the padded gather, the independent-clips mode of BASELINE
config 5 -- is the product's code.  ONE spawn of eight workers runs every scenario in sequence (a spawn costs a torch import per rank);
frames are compared through SHA-1 digests so that nothing large crosses the result queue."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 8
H, W, R, C, OMEGA, ITERS = 24, 32, 2, 2, 3, 6
# (frames of the whole clip, gather?)  300 = 38 x 7 + 34; 12 = two empty shards; 5 < ranks: three empty; 2400 = config 4's 8 x 300
SCENARIOS = [(300, False), (300, True), (12, False), (12, True), (5, True), (2400, False), (2400, True)]


def _clip(F, seed):
    sys.path.insert(0, REPO)
    from meshflow_amd import synthetic
    return synthetic.clip(F, H, W, R, C, seed=seed, kind='noise', jitter_sigma=1.0)


def _stages(frames, disp, hom):
    from oracle import clib, meshflow_oracle as mo

    def jacobi_fn():           # replicated on every rank: identical bits, no communication
        return torch.from_numpy(mo.stabilized_vertex_displacements(W, H, 0, disp, hom, OMEGA, ITERS))

    def warp_fn(lo, hi, stab_all):
        if hi == lo:           # an empty shard (F < ranks * ceil(F / ranks)): nothing to warp, nothing to contribute
            return torch.empty((0, H, W, 3), dtype=torch.uint8), torch.empty((0, 4), dtype=torch.int32)
        out, crop, bad = clib.warp_clip(frames[lo:hi], R, C, disp[lo:hi], stab_all[lo:hi].numpy())
        assert bad == 0
        return torch.from_numpy(out), torch.from_numpy(crop)

    def crop_reduce_fn(crop):  # mfs.py:992-995: the defaults are the neutral element of mfs.py:1103-1106's max / min
        if crop.shape[0] == 0:
            return torch.tensor([0, 0, W - 1, H - 1], dtype=torch.int32)
        return torch.stack([crop[:, 0].max(), crop[:, 1].max(), crop[:, 2].min(), crop[:, 3].min()]).to(torch.int32)

    return jacobi_fn, warp_fn, crop_reduce_fn


def _digest(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def _worker(rank, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      OMP_NUM_THREADS='1')
    torch.set_num_threads(1)
    from meshflow_amd import dist as mfdist, host
    r, w, device = mfdist.init_from_env('cpu')
    assert (r, w) == (rank, WORLD) and device.type == 'cpu'
    res = {'rank': rank}
    for F, gather in SCENARIOS:
        frames, disp, hom = _clip(F, seed=4)
        out, bounds, stab_all, (lo, hi) = mfdist.stabilize_sharded(F, *_stages(frames, disp, hom), gather=gather)
        assert (lo, hi) == host.shard_range(F, WORLD, rank)
        if gather:
            assert (out is None) == (rank != 0)
            res[(F, gather)] = {'bounds': bounds.tolist(), 'lo': lo, 'hi': hi, 'n': None if out is None else out.shape[0],
                                'digest': None if out is None else _digest(out.numpy())}
        else:
            assert out.shape[0] == hi - lo
            res[(F, gather)] = {'bounds': bounds.tolist(), 'lo': lo, 'hi': hi, 'n': hi - lo,
                                'digests': [_digest(f) for f in out.numpy()]}
    # BASELINE config 5: eight independent clips, one per rank (seed = rank), no collective in the data path -- the shard is the whole
    # clip and the rectangle is the rank's own
    F = 20
    frames, disp, hom = _clip(F, seed=rank)
    out, bounds, _, (lo, hi) = mfdist.stabilize_sharded(F, *_stages(frames, disp, hom), shard=(1, 0), collective=False)
    assert (lo, hi) == (0, F)
    res['clips'] = {'bounds': bounds.tolist(), 'digest': _digest(out.numpy())}
    # the scalar the bench takes its time from: MAX over ranks of a float64
    t = torch.tensor([float(rank)], dtype=torch.float64)
    res['max'] = float(mfdist.all_reduce_max(t).item())
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def _single(F, seed):
    frames, disp, hom = _clip(F, seed)
    jacobi_fn, warp_fn, crop_reduce_fn = _stages(frames, disp, hom)
    out, crop = warp_fn(0, F, jacobi_fn())
    return out.numpy(), crop_reduce_fn(crop).tolist()


@pytest.fixture(scope='module')
def eight_ranks():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    try:
        results = sorted((q.get(timeout=240) for _ in range(WORLD)), key=lambda d: d['rank'])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()                    # (exact process objects we started)
    assert [p.exitcode for p in procs] == [0] * WORLD
    return results


def test_partition_covers_every_frame_once():
    sys.path.insert(0, REPO)
    from meshflow_amd import host
    for F in (1, 5, 8, 12, 300, 2399, 2400, 2401):
        ranges = [host.shard_range(F, WORLD, g) for g in range(WORLD)]
        assert ranges[0][0] == 0 and ranges[-1][1] == F
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and all(lo <= hi for lo, hi in ranges)
    assert [hi - lo for lo, hi in (host.shard_range(300, WORLD, g) for g in range(WORLD))] == [38] * 7 + [34]
    assert [hi - lo for lo, hi in (host.shard_range(12, WORLD, g) for g in range(WORLD))] == [2] * 6 + [0, 0]


@pytest.mark.parametrize('F,gather', SCENARIOS)
def test_eight_rank_pass_equals_single_process(eight_ranks, F, gather):
    want, want_bounds = _single(F, seed=4)
    per_rank = [r[(F, gather)] for r in eight_ranks]
    for res in per_rank:
        assert res['bounds'] == want_bounds                          # every rank holds the CLIP-level rectangle (mfs.py:1103-1106)
    assert all(a['hi'] == b['lo'] for a, b in zip(per_rank, per_rank[1:]))
    if gather:
        assert per_rank[0]['n'] == F and per_rank[0]['digest'] == _digest(want)      # one gather, padding trimmed, frames in order
        assert all(res['digest'] is None for res in per_rank[1:])
    else:
        got = [d for res in per_rank for d in res['digests']]
        assert got == [_digest(f) for f in want]


def test_eight_independent_clips_and_max_over_ranks(eight_ranks):
    for res in eight_ranks:
        want, want_bounds = _single(20, seed=res['rank'])
        assert res['clips']['digest'] == _digest(want) and res['clips']['bounds'] == want_bounds     # its own clip, its own rectangle
        assert res['max'] == float(WORLD - 1)
    assert len({res['clips']['digest'] for res in eight_ranks}) == WORLD                             # (the clips do differ)
