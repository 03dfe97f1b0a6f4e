"""bench.py --gpus N without a launcher: the parent refuses cleanly (exit code 2, no JSON) when the box has too few GPUs --
here: none -- before starting any child; and its launcher with EIGHT stand-in ranks (tests/fake_rank.py) under gloo: environment,
free port, rank 0's single line, and what happens when a rank dies or hangs."""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_without_gpus_exits_2():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--workload', 'small'], cwd=REPO,
                          capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.device_count() >= 2:
        return                                # a real multi-GPU box: covered by the -m gpu tests
    assert proc.returncode == 2 and 'GPU(s) visible' in proc.stderr and not proc.stdout.strip()


def test_scale_all_dry_run_commands_parse():
    """tools/scale_all.sh --dry-run prints every command of the north-star sweep; each must be accepted by bench.py's own argument
    parser (tools/capi_shard_run.py's by its own), name a known workload, and together they must cover N = 1, 2, 4, 8 of config 2,
    config 4 = cfg4shard x N, config 5 = --mode clips and the host-to-host mode -- so that the first multi-GPU lease cannot fail on a flag."""
    import shlex
    sys.path.insert(0, REPO)
    import bench
    proc = subprocess.run(['bash', os.path.join(REPO, 'tools', 'scale_all.sh'), '--dry-run'], cwd=REPO, capture_output=True, text=True, timeout=60)
    assert proc.returncode == 0, proc.stderr
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) >= 14
    parser = bench.build_parser()
    seen = set()
    for line in lines:
        words = shlex.split(line)
        assert words[0] == 'python'
        if words[1] == 'bench.py':
            a = parser.parse_args(words[2:])               # SystemExit(2) on an unknown flag or a bad choice
            assert a.workload in bench.WORKLOADS and a.gpus in (1, 2, 4, 8) and a.steps > 0
            seen.add((a.workload, a.mode, a.gpus))
        else:
            assert words[1] == 'tools/capi_shard_run.py' and os.path.exists(os.path.join(REPO, words[1])) and words[2:] == ['--gpus', '8']
    for n in (1, 2, 4, 8):
        assert ('cfg2', 'shard', n) in seen
    for n in (2, 4, 8):
        assert ('cfg4shard', 'shard', n) in seen and ('cfg2', 'clips', n) in seen and ('cfg2', 'e2e', n) in seen
    # and the launcher's own default run: flags of the driver's command line
    a = parser.parse_args(['--gpus', '8', '--steps', '20', '--warmup', '5'])
    assert (a.gpus, a.steps, a.warmup, a.workload, a.mode) == (8, 20, 5, 'cfg2', 'shard')


_LAUNCH = """
import sys, types
sys.path.insert(0, {repo!r})
import bench
raise SystemExit(bench.launch_children(types.SimpleNamespace(gpus=8), script={script!r}, argv={argv!r}, check_gpus=False))
"""


def _launch8(argv, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['MESHFLOW_DIST_BACKEND'] = 'gloo'
    code = _LAUNCH.format(repo=REPO, script=os.path.join(REPO, 'tests', 'fake_rank.py'), argv=argv)
    t0 = time.monotonic()
    proc = subprocess.run([sys.executable, '-c', code], cwd=REPO, capture_output=True, text=True, timeout=timeout, env=env)
    return proc, time.monotonic() - t0


def test_launcher_eight_ranks_one_line():
    """Eight children with torchrun's environment on a port the launcher picked: every collective bench.py's timing uses completes, the
    ragged partition (300 = 38 x 7 + 34) sums to the clip, rank 0 alone prints."""
    proc, _ = _launch8(['--frames', '300'])
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip() and not l.startswith('[Gloo]')]       # (gloo announces its peers on stdout)
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {'n_gpus': 8, 'elapsed': 0.008, 'frames': 300, 'bounds': [7, 14, 93, 36]}


def test_launcher_ends_the_others_when_a_rank_dies():
    """Rank 5 exits with code 3 before the first barrier; rank 6 ignores SIGTERM.  The launcher returns 3, has terminated the
    stranded ranks and killed the deaf one -- within its 10 s grace period, not after gloo's timeout."""
    proc, took = _launch8(['--die', '5', '3', '--hang', '6'])
    assert proc.returncode == 3 and not [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert took < 60
