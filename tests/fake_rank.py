"""Stand-in for one rank of `bench.py --gpus N` on a box without GPUs (tests/test_bench_launcher.py): the launcher's environment ->
meshflow_amd.dist.init_from_env('cpu') -> the collectives bench.py's timing rests on (barrier, MAX over ranks) -> rank 0 prints ONE
JSON line.  `--die RANK CODE`: that rank exits with CODE before the first collective (the others are then stuck in the barrier and the
launcher must end them); `--hang RANK`: that rank ignores SIGTERM as well (the launcher must kill it)."""
import json
import os
import signal
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    argv = sys.argv[1:]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['LOCAL_RANK']) == rank and os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    if '--die' in argv and rank == int(argv[argv.index('--die') + 1]):
        sys.exit(int(argv[argv.index('--die') + 2]))
    if '--hang' in argv and rank == int(argv[argv.index('--hang') + 1]):
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
    import torch
    import torch.distributed as dist
    from meshflow_amd import dist as mfdist, host
    r, w, device = mfdist.init_from_env('cpu')
    assert (r, w) == (rank, world)
    dist.barrier()
    F = int(argv[argv.index('--frames') + 1]) if '--frames' in argv else 300
    lo, hi = host.shard_range(F, world, rank)
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    elapsed = float(mfdist.all_reduce_max(t).item())
    count = torch.tensor([float(hi - lo)], dtype=torch.float64)
    dist.all_reduce(count)
    bounds = mfdist.allreduce_crop(torch.tensor([rank, 2 * rank, 100 - rank, 50 - 2 * rank], dtype=torch.int32))
    if rank == 0:
        print(json.dumps({'n_gpus': world, 'elapsed': elapsed, 'frames': int(count.item()), 'bounds': bounds.tolist()}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
