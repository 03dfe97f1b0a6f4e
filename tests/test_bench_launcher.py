"""bench.py --gpus N without a launcher: the parent refuses cleanly (exit code 2, no JSON) when the box has too few GPUs --
here: none -- before starting any child."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_without_gpus_exits_2():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    proc = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--workload', 'small'], cwd=REPO,
                          capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.device_count() >= 2:
        return                                # a real multi-GPU box: covered by the -m gpu tests
    assert proc.returncode == 2 and 'GPU(s) visible' in proc.stderr and not proc.stdout.strip()
