# Every mode / flag of bench.py once on the small workload (N = 1, and N = 2, 3 under gloo on one GPU): each must print its JSON line.
#   gpurun -- bash tools/bench_flags.sh
cd $GRAFT_REPO_ROOT
run() { echo -n "bench.py $* :: "; out=$(timeout 600 python bench.py "$@" 2>/tmp/bench_err.txt); rc=$?; if [ $rc -ne 0 ]; then echo "rc=$rc"; tail -3 /tmp/bench_err.txt; else echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ok', round(d['value']), d['n_gpus'], d.get('metric','')[:40])"; fi; }
B="--workload small --steps 3 --warmup 1 --cpu-frames 0"
run $B
run $B --no-e2e --no-workloads
run $B --pipeline serial
run $B --chunks 4
run $B --chunks 1 --rectangle early
run $B --rectangle early
run $B --frames 37
run $B --as-rank-of 8
run $B --as-rank-of 8 --frames 10
run $B --checksum
run $B --mode clips
run $B --mode e2e
run --workload small --steps 2 --warmup 1 --cpu-frames 4 --no-faithful
export MESHFLOW_DIST_BACKEND=gloo
run $B --gpus 2
run $B --gpus 2 --mode clips
run $B --gpus 2 --mode e2e
run $B --gpus 2 --no-gather
run $B --gpus 2 --checksum
run $B --gpus 2 --rectangle early
run $B --gpus 2 --chunks 2
run $B --gpus 3 --frames 5
