"""The multi-GPU orchestration (meshflow_amd/dist.py) under gloo with world_size 2 on CPU.

The HIP kernels cannot run here, so the oracle stands in for them: the test exercises the REAL sharding,
crop all-reduce and frame gather code paths and checks that the sharded result equals the single-process
result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, F, gather, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from meshflow_amd import dist as mfdist, host, synthetic
    from oracle import clib, meshflow_oracle as mo
    r, w, device = mfdist.init_from_env('cpu')
    assert (r, w) == (rank, world) and device.type == 'cpu'
    H, W, R, C, omega, iters = 48, 64, 4, 4, 3, 8
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=4, kind='noise', jitter_sigma=1.0)

    def jacobi_fn():           # replicated on every rank
        return torch.from_numpy(mo.stabilized_vertex_displacements(W, H, 0, disp, hom, omega, iters))

    def warp_fn(lo, hi, stab_all):
        out, crop, bad = clib.warp_clip(frames[lo:hi], R, C, disp[lo:hi], stab_all[lo:hi].numpy())
        assert bad == 0
        return torch.from_numpy(out), torch.from_numpy(crop)

    def crop_reduce_fn(crop):
        if crop.shape[0] == 0:
            return torch.tensor([0, 0, W - 1, H - 1], dtype=torch.int32)
        return torch.stack([crop[:, 0].max(), crop[:, 1].max(), crop[:, 2].min(), crop[:, 3].min()]).to(torch.int32)

    entered = []

    class _Ctx:                 # stands in for the HIP pipeline's prep-stream context (the crop all-reduce is issued under it)
        def __enter__(self):
            entered.append(1)

        def __exit__(self, *exc):
            return False

    if F == 9:
        # the way MeshFlowStabilizer.stabilize_resident calls it (round 5): the caller names the frames it holds (`frame_range`, here an
        # UNEVEN split the default partition would not make), its warp stage hands back the shard's rectangle itself (the kernels fold it
        # together) and the reduction stage is the identity
        mine = (0, 6) if rank == 0 else (6, 9)
        out, bounds, stab_all, (lo, hi) = mfdist.stabilize_sharded(
            F, jacobi_fn, lambda lo_, hi_, st_: (lambda o_c: (o_c[0], crop_reduce_fn(o_c[1])))(warp_fn(lo_, hi_, st_)), lambda b: b,
            gather=False, frame_range=mine, collective=True)
        assert (lo, hi) == mine
    else:
        out, bounds, stab_all, (lo, hi) = mfdist.stabilize_sharded(F, jacobi_fn, warp_fn, crop_reduce_fn, gather=gather,
                                                                 exchange_ctx=_Ctx if F % 2 else None)
        assert len(entered) == (1 if F % 2 else 0)
        assert (lo, hi) == host.shard_range(F, world, rank)
    res = {'rank': rank, 'bounds': bounds.tolist(), 'lo': lo, 'hi': hi}
    if gather:
        res['frames'] = None if out is None else out.numpy()
    else:
        res['frames'] = out.numpy()
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def _single(F):
    sys.path.insert(0, REPO)
    from meshflow_amd import synthetic
    from oracle import clib, meshflow_oracle as mo
    H, W, R, C, omega, iters = 48, 64, 4, 4, 3, 8
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=4, kind='noise', jitter_sigma=1.0)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, omega, iters)
    out, crop, _ = clib.warp_clip(frames, R, C, disp, stab)
    return out, [int(crop[:, 0].max()), int(crop[:, 1].max()), int(crop[:, 2].min()), int(crop[:, 3].min())]


@pytest.mark.parametrize('F,gather', [(10, False), (7, True), (1, True), (9, False)])
def test_sharded_pass_equals_single_process(F, gather):
    world = 2
    port = 29600 + (os.getpid() + F) % 300
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=180) for _ in range(world)), key=lambda d: d['rank'])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want, want_bounds = _single(F)
    for res in results:
        assert res['bounds'] == want_bounds                       # every rank holds the clip-level crop
    if gather:
        assert results[1]['frames'] is None
        np.testing.assert_array_equal(results[0]['frames'], want)      # one gather, padded shards trimmed
    else:
        got = np.concatenate([r['frames'] for r in results])
        np.testing.assert_array_equal(got, want)
        assert results[0]['hi'] == results[1]['lo']
