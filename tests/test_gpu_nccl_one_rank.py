"""torch.distributed's "nccl" backend (= RCCL) on the test box's ONE GPU: a process group of one rank still runs the collectives, so the
product's exchange steps -- the 16-byte MAX all-reduce of the crop rectangle inside `stabilize_resident(collective=True)` (on the caller's
stream, and on the prep stream with `resident_rectangle = 'early'`) and the padded frame gather of `dist.gather_frames` -- execute on RCCL
with the tensors, dtypes and streams a real N-GPU job hands them.  (What more than one rank adds -- xGMI, the rendezvous of several
processes -- needs a node; `tests/test_dist_gloo.py` covers the N = 2 logic under gloo.)"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from meshflow_amd import dist as mfdist, ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    rank, world, dev = mfdist.init_from_env('cuda')
    assert (rank, world) == (0, 1) and not dist.is_initialized()           # init_from_env leaves a single process alone ...
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)    # ... the group of one is this test's doing
    assert mfdist.active() and mfdist.world_size() == 1 and dist.get_backend() == 'nccl'
    F, H, W, R, C = 40, 360, 640, 8, 8
    disp, hom = synthetic.motion(F, R, C, seed=5)
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=5)
    d_disp = torch.from_numpy(disp).to(dev)
    want = None
    for rectangle in ('late', 'early'):
        s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=5, optimization_num_iterations=20, device='cuda:0')
        s.resident_rectangle = rectangle
        plain, b0, stab0 = s.stabilize_resident(d_frames, d_disp, hom, collective=False)
        outs = [s.stabilize_resident(d_frames, d_disp, hom, collective=True, check='deferred') for _ in range(3)]        # back to back: no synchronisation
        s.finish()
        torch.cuda.synchronize()
        for frames, bounds, stab in outs:
            assert torch.equal(frames, plain) and torch.equal(stab, stab0)
            assert bounds.tolist() == b0.tolist(), (rectangle, bounds.tolist(), b0.tolist())
        want = b0.tolist() if want is None else want
        assert b0.tolist() == want
    gathered = mfdist.gather_frames(plain, F)
    torch.cuda.synchronize()
    assert gathered.shape == plain.shape and torch.equal(gathered, plain)
    ragged = mfdist.gather_frames(plain[:37], 37)
    assert torch.equal(ragged, plain[:37])
    t = torch.tensor([3, -7, 9], dtype=torch.int32, device=dev)
    assert mfdist.all_reduce_max(t).tolist() == [3, -7, 9]
    secs = torch.tensor([0.00125], dtype=torch.float64, device=dev)            # bench.py's max-over-ranks of the elapsed time
    assert float(mfdist.all_reduce_max(secs).item()) == 0.00125
    dist.barrier()
    dist.destroy_process_group()
    print('NCCL-ONE-RANK-OK', want)
''') % REPO


def test_exchange_steps_on_rccl_with_one_rank():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MESHFLOW_DIST_BACKEND')}
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29700 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY='0')
    proc = subprocess.run([sys.executable, '-c', SCRIPT], cwd=REPO, capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, (proc.stdout[-1500:], proc.stderr[-3000:])
    assert 'NCCL-ONE-RANK-OK' in proc.stdout
