"""-m gpu, ONE GPU: everything of the N-GPU path that can be exercised without N GPUs.

  * `bench.py --as-rank-of 8` (what rank 0 of an 8-GPU job does) produces exactly frames [lo, hi) of the whole-clip run;
  * a RAGGED split (300 frames over 8 ranks: 38, 38, ... , 34) through dist.stabilize_sharded with the HIP operators, rank by
    rank, reassembles the whole-clip result and its crop rectangle;
  * `bench.py --gpus 2` under gloo: the GATHERED frames on rank 0 equal one process warping the whole clip;
  * `bench.py --gpus 2 --mode clips` (BASELINE config 5: independent clips, no collective).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return torch.device('cuda:0')


def _bench(*flags, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    e.update(env or {})
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), *flags], cwd=REPO, capture_output=True, text=True,
                          timeout=1500, env=e)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def _whole_clip(dev, workload, F, seed=0):
    """One process, the whole clip of F frames device-resident: per-frame checksums and clip-level crop bounds."""
    sys.path.insert(0, REPO)
    import bench
    from meshflow_amd import ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    H, W, _, R, C, omega, iters = bench.WORKLOADS[workload]
    disp, hom = synthetic.motion(F, R, C, seed=seed)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters, device='cuda:0')
    d_disp = torch.from_numpy(disp).to(dev)
    d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
    sums, crops = [], []
    step = 150 if H < 2000 else 50                                   # bounded device memory: warp the clip in slices
    for i0 in range(0, F, step):
        i1 = min(F, i0 + step)
        frames = synthetic.frames_torch(i1 - i0, H, W, dev, seed=seed, first_frame=i0)
        out, crop = s._stabilized_frames_device(frames, d_disp[i0:i1], d_stab[i0:i1])
        sums += bench.frame_checksums(out)
        crops.append(crop.clone())
        del frames, out
    return sums, ops.crop_reduce(torch.cat(crops), W, H).tolist()


@pytest.mark.parametrize('workload,frames', [('cfg2', 300), ('cfg4shard', 150)])
def test_as_rank_of_8_equals_its_slice_of_the_whole_clip(dev, workload, frames):
    """Rank 0 of an 8-GPU job on ONE GPU (clip of 8 x frames, replicated Jacobi over all of them, own frame range): its stabilized
    frames are frames [lo, hi) of the whole clip."""
    ranks = 8 if workload == 'cfg2' else 2               # (the 4K whole-clip reference is kept to 2 shards: 300 frames of 3840x2160)
    d = _bench('--workload', workload, '--as-rank-of', str(ranks), '--steps', '1', '--warmup', '0', '--cpu-frames', '0', '--no-e2e', '--checksum')
    lo, hi = d['frames_checksum_range']
    assert (lo, hi) == (0, frames) and d['as_rank_of'] == ranks
    want, _ = _whole_clip(dev, workload, frames * ranks)
    assert d['frames_checksum'] == want[lo:hi]


def test_ragged_split_300_frames_over_8_ranks_with_the_hip_operators(dev):
    """dist.stabilize_sharded rank by rank with shard=(8, g): 300 frames do not divide by 8 (38 x 7 + 34); the shards reassemble the
    whole-clip frames and the all-reduce of their rectangles is the whole clip's."""
    from meshflow_amd import dist as mfdist, host, ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    H, W, F, R, C = 360, 640, 300, 16, 16
    disp, hom = synthetic.motion(F, R, C, seed=3)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, device='cuda:0')
    d_disp = torch.from_numpy(disp).to(dev)
    all_frames = synthetic.frames_torch(F, H, W, dev, seed=3)
    d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
    whole, whole_crop = s._stabilized_frames_device(all_frames, d_disp, d_stab)
    whole_bounds = ops.crop_reduce(whole_crop, W, H)
    sizes, parts, rects = [], [], []
    for g in range(8):
        def warp_fn(lo, hi, stab_all):
            return s._stabilized_frames_device(all_frames[lo:hi], d_disp[lo:hi], stab_all[lo:hi])
        frames, bounds, stab_all, (lo, hi) = mfdist.stabilize_sharded(
            F, lambda: s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom), warp_fn,
            lambda crop: ops.crop_reduce(crop, W, H), shard=(8, g), collective=False)
        assert (lo, hi) == host.shard_range(F, 8, g) and torch.equal(stab_all, d_stab)
        sizes.append(hi - lo); parts.append(frames); rects.append(bounds)
    assert sizes == [38] * 7 + [34]
    assert torch.equal(torch.cat(parts), whole)
    r = torch.stack(rects)
    merged = torch.stack([r[:, 0].max(), r[:, 1].max(), r[:, 2].min(), r[:, 3].min()])
    assert torch.equal(merged, whole_bounds)


def test_two_rank_bench_gathers_the_whole_clip(dev):
    """`python bench.py --gpus 2` (gloo: both ranks on this GPU): the frames GATHERED on rank 0 equal one process warping all of them."""
    d = _bench('--workload', 'small', '--gpus', '2', '--steps', '1', '--warmup', '0', '--checksum', env={'MESHFLOW_DIST_BACKEND': 'gloo'})
    assert d['frames_checksum_range'] == [0, 128] and d['gather_to_rank0_ms'] > 0
    want, bounds = _whole_clip(dev, 'small', 128)
    assert d['frames_checksum'] == want
    assert d['crop_bounds'] == bounds


def test_two_rank_bench_clips_mode(dev):
    """BASELINE config 5 in miniature: independent clips, one per rank (seed = rank), no collective, no gather."""
    d = _bench('--workload', 'small', '--gpus', '2', '--mode', 'clips', '--steps', '2', '--warmup', '1', env={'MESHFLOW_DIST_BACKEND': 'gloo'})
    assert d['n_gpus'] == 2 and 'independent clips' in d['config']['parallelism'] and 'gather_to_rank0_ms' not in d
    assert abs(d['value'] - 2 * 64 * 2 / (d['ms_per_step'] * 2e-3)) < 1e-6 * d['value']      # both ranks' frames count
    want, bounds = _whole_clip(dev, 'small', 64, seed=0)                                     # rank 0's clip (seed 0)
    assert d['crop_bounds'] == bounds
