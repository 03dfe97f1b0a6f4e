"""bench.py prints ONE JSON line with the fields the driver's contract names (run on the small workload); `--gpus 2` by
itself starts two ranks and runs the real sharded step, all-reduce and gather of bench.py (gloo, both ranks on one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--workload', 'small', *flags], cwd=REPO,
                          capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def _contract(d, n_gpus, steps, warmup):
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == n_gpus and d['steps'] == steps and d['warmup'] == warmup and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['unit'] == 'frames/s' and d['value'] > 0
    assert 'workload' in d['config'] and 'model' not in d['config']


def test_bench_json_line_contract():
    d = _run('--steps', '3', '--warmup', '1', '--cpu-frames', '4')
    _contract(d, 1, 3, 1)
    r = d['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    c = d['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in c, key
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0
    f = c['reference_faithful']                                  # the reference's own formulation, scaled from a sample
    assert f['kind'].startswith('reference-faithful, scaled from') and 0 < f['value'] < c['value']
    e = d['end_to_end']
    assert e['value'] > 0 and e['runs'] >= 5 and e['min_ms_per_clip'] <= e['ms_per_clip']
    # BASELINE's host-to-host figure is a timed region of its own: steps x ms_per_step = the seconds it took
    assert e['steps'] == e['runs'] and abs(e['steps'] * e['ms_per_step'] * 1e-3 - e['elapsed_s']) < 1e-6 * e['elapsed_s'] + 1e-9
    assert e['with_crop']['steps'] == e['with_crop']['runs'] and e['with_crop']['value'] > 0
    k = d['kernel_path']                                          # the HBM-resident figure = `value`, through the public method
    assert k['value'] == d['value'] and k['ms_per_step'] == d['ms_per_step'] and 'stabilize_resident' in k['through']
    assert k['serial']['ms_per_step'] > 0 and k['latency_ms_single_clip']['median'] > 0
    assert 'workloads' not in d                                   # only beside the default workload (below)
    assert d['next_rows']['vertex_motion']['avg_ms'] > 0
    assert d['cfg1'].startswith('skipped')                       # BASELINE configs[0]: no decoder / no video on the box
    assert d['communicator']['world_size'] == 1


def test_bench_default_run_carries_the_other_configurations():
    """The driver's own command line (default workload = BASELINE configs[1]): the line also measures configs 3 and 4-shard in the same
    run -- warp launch time and roofline fraction, step time, Jacobi kernel, host-to-host clip against the PCIe rate of this run."""
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '5', '--warmup', '2', '--cpu-frames', '8',
                           '--no-faithful', '--workload-steps', '4'], cwd=REPO, capture_output=True, text=True, timeout=1500)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _contract(d, 1, 5, 2)
    assert d['config']['workload'].startswith('cfg2: 1920x1080, 300 frames')
    assert 0.2 < d['roofline']['frac'] < 1.0 and d['end_to_end']['with_crop']['value'] > 500          # north_star: >= 500 frames/s end to end
    # the figure north_star's 500 frames/s is quoted on sits in the fields the driver keeps, next to the kernel's fractions of both ceilings
    assert d['config']['end_to_end_fps'] == d['end_to_end']['with_crop']['value'] == d['roofline']['end_to_end_fps']
    r = d['roofline']
    assert r['peak_achievable'] == 6290.0 and abs(r['frac_of_achievable'] - r['achieved'] / r['peak_achievable']) < 1e-9 and r['event_avg_launch_ms'] == r['avg_launch_ms']
    for key in ('', 'with_crop', 'pinned_buffers'):
        e = (d['end_to_end'][key] if key else d['end_to_end'])['roofline']
        assert e['peak'] == 63.0 and 0.15 < e['frac'] <= 1.0 and abs(e['frac'] - e['achieved'] / 63.0) < 1e-9
    if 'trace' in r:                                                   # the last kept rocprofv3 trace average, beside the event time
        assert r['trace']['avg_launch_ms'] > 0 and 0.2 < r['trace']['frac'] < 1.0 and r['trace']['source'].startswith('profiles/')
    for name, frames in (('cfg3', 600), ('cfg4shard', 150)):
        w = d['workloads'][name]
        assert 'error' not in w, w
        assert w['steps'] == 4 and w['ms_per_step'] > 0 and abs(w['value'] - frames / (w['ms_per_step'] * 1e-3)) < 1e-6 * w['value']
        assert 0.2 < w['warp']['frac'] < 1.0 and w['warp']['avg_launch_ms'] < w['ms_per_step']
        # the host path's ceiling is the LINK (PCIe Gen5 x16, 63 GB/s per direction): a fraction of it can never exceed 1; the probes of this
        # box's copy rates are context (ADVICE r5 / VERDICT r5 weak 5: against a probe the "fraction" came out above 1)
        e = w['end_to_end']['roofline']
        assert w['jacobi']['kernel_ms'] > 0 and w['end_to_end']['value'] > 0 and e['peak'] == 63.0 and 0.15 < e['frac'] <= 1.0
        assert abs(e['frac'] - e['achieved'] / e['peak']) < 1e-9 and e['achieved'] > 10.0            # (absolute: > 10 GB/s each way)
        assert 0.25 < w['warp']['frac_of_achievable'] < 1.0 and abs(w['warp']['frac_of_achievable'] / w['warp']['frac'] - 8.0 / 6.29) < 1e-6


def test_bench_e2e_mode_value_is_the_host_to_host_clip():
    d = _run('--mode', 'e2e', '--steps', '3', '--warmup', '1')
    _contract(d, 1, 3, 1)
    assert 'end-to-end' in d['metric'] and d['min_ms_per_step_rank0'] <= d['ms_per_step'] * 1.001
    assert abs(d['value'] - 64 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']


def test_bench_two_ranks_started_by_bench_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts both ranks.  gloo so that they can share the one
    GPU of the test box; the step, the crop all-reduce, the frame gather and the sharded drain are bench.py's own code."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['MESHFLOW_DIST_BACKEND'] = 'gloo'
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--workload', 'small', '--gpus', '2', '--steps', '2',
                           '--warmup', '1'], cwd=REPO, capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = json.loads(lines[0])
    _contract(d, 2, 2, 1)
    assert d['communicator'] == {**d['communicator'], 'world_size': 2, 'backend': 'gloo'}
    assert d['gather_to_rank0_ms'] > 0 and d['sharded_d2h_ms'] > 0 and 'gather_error' not in d
    assert '128 total' in d['config']['workload'] and d['cpu_baseline'] is None
    # the clip-level crop bounds of the two-rank run = those of ONE process warping all 128 frames of the same clip
    from meshflow_amd import ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    import torch
    dev = torch.device('cuda:0')
    disp, hom = synthetic.motion(128, 16, 16, seed=0)
    s = MeshFlowStabilizer(device='cuda:0')
    d_disp = torch.from_numpy(disp).to(dev)
    d_stab = s._stabilized_vertex_displacements_device(d_disp, 640, 360, 0, hom)
    _, crop = s._stabilized_frames_device(synthetic.frames_torch(128, 360, 640, dev, seed=0), d_disp, d_stab)
    assert d['crop_bounds'] == ops.crop_reduce(crop, 640, 360).tolist()


def test_bench_two_ranks_under_the_drivers_own_launcher():
    """The driver's command line for N > 1, word for word: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` (gloo, so that the two ranks can share the test box's one GPU): rank 0
    prints the ONE JSON line, the other rank nothing."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['MESHFLOW_DIST_BACKEND'] = 'gloo'
    port = 29500 + os.getpid() % 400
    proc = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                           '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--workload', 'small', '--gpus', '2', '--steps', '2',
                           '--warmup', '1'], cwd=REPO, capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = json.loads(lines[0])
    _contract(d, 2, 2, 1)
    assert d['communicator']['world_size'] == 2 and d['communicator']['backend'] == 'gloo'
    assert d['gather_to_rank0_ms'] > 0 and 'gather_error' not in d and '128 total' in d['config']['workload']


def test_bench_failed_rank_gives_nonzero_exit():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['MESHFLOW_DIST_BACKEND'] = 'no-such-backend'
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--workload', 'small', '--gpus', '2', '--steps', '1',
                           '--warmup', '0'], cwd=REPO, capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode != 0
    assert not [l for l in proc.stdout.splitlines() if l.strip().startswith('{')]


def test_bench_eight_ranks_under_the_drivers_own_launcher():
    """The SCALE run's shape at N = 8 on the one-GPU test box: the driver's launcher line with eight gloo ranks sharing the GPU, `bench.py`'s
    own step (frame-range shard, replicated sweep over all 8 x 6 frames, 16-byte all-reduce), gather and sharded drain.  The gathered clip's
    per-frame checksums and the clip rectangle equal ONE process doing the whole 48-frame clip."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['MESHFLOW_DIST_BACKEND'] = 'gloo'
    port = 29900 + os.getpid() % 90
    common = ['--workload', 'small', '--steps', '2', '--warmup', '1', '--cpu-frames', '0', '--checksum']
    proc = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                           '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '8', '--frames', '6'] + common,
                          cwd=REPO, capture_output=True, text=True, timeout=1500, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = json.loads(lines[0])
    _contract(d, 8, 2, 1)
    assert d['communicator']['world_size'] == 8 and d['communicator']['backend'] == 'gloo' and '48 total' in d['config']['workload']
    assert d['gather_to_rank0_ms'] > 0 and 'gather_error' not in d and d['frames_checksum_range'] == [0, 48] and len(d['frames_checksum']) == 48
    one = _run('--frames', '48', '--no-e2e', '--no-workloads', *common[2:])
    assert one['frames_checksum'] == d['frames_checksum'] and one['crop_bounds'] == d['crop_bounds']
