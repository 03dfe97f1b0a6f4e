#!/bin/bash
# The whole north-star table in ONE command, for the first box that has more than one GPU (nothing here has ever run on more than one):
#
#     tools/scale_all.sh [--dry-run] [MAXGPUS] [OUTDIR]          default: every GPU of the node, gpurun_out/scale
#
# --dry-run prints every command of the sweep (one per line, nothing runs, no GPU needed): tests/test_bench_launcher.py parses them with
# bench.py's own argument parser, so that the first 8-GPU lease fails on RCCL itself, if at all, not on a flag.
#
#   cfg2, frame-range shards, N = 1, 2, 4, 8     (weak scaling: 300 frames per GPU, Jacobi replicated, 16-byte crop all-reduce, final gather)
#   cfg4 = cfg4shard x N                          (3840x2160, 150 frames per GPU: BASELINE config 4 at N = 8)
#   cfg5 = --mode clips                           (N independent cfg2 clips, no collective)
#   host-to-host = --mode e2e                     (N independent clips over N PCIe links)
#   the C ABI's own exchange path                 (tools/capi_shard_run.py: ONE process, mf_comm_init_all / mf_allreduce_crop / mf_gather_frames)
#
# Every bench line carries communicator.world_size, gather_to_rank0_ms and sharded_d2h_ms; lines are appended to OUTDIR/scale.jsonl with
# the command in front, stderr (RCCL version, free memory before the gather) to OUTDIR/*.err.  A failing run is recorded and the sweep goes on.
set -u
cd "$(dirname "$0")/.."
DRY=0
if [ "${1:-}" = "--dry-run" ]; then DRY=1; shift; fi
# (kept on purpose: the host driver of this pool only supports dmabuf IPC -- without it RCCL and cross-process tensor sharing fail with
# "hipIpcGetMemHandle: invalid argument"; see the Environment notes of the build)
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
if [ "$DRY" = 1 ]; then NGPU=8; else NGPU=$(python3 -c "import bench; print(bench.visible_gpus() or 1)"); fi
MAX=${1:-$NGPU}
OUT=${2:-gpurun_out/scale}
if [ "$DRY" = 0 ]; then
    mkdir -p "$OUT"
    : > "$OUT/scale.jsonl"
fi
run() {     # name, then the bench flags
    local name=$1; shift
    if [ "$DRY" = 1 ]; then echo "python bench.py $*"; return; fi
    echo "== $name: python bench.py $*" | tee -a "$OUT/scale.jsonl" >&2
    if timeout 1500 python bench.py "$@" >> "$OUT/scale.jsonl" 2> "$OUT/$name.err"; then :; else
        echo "{\"failed\": \"$name\", \"rc\": $?}" >> "$OUT/scale.jsonl"
    fi
}
for n in 1 2 4 8; do
    [ "$n" -le "$MAX" ] || continue
    run "cfg2_shard_n$n" --gpus $n --steps 20 --warmup 5 --cpu-frames 0 --no-e2e --no-workloads
done
for n in 2 4 8; do
    [ "$n" -le "$MAX" ] || continue
    run "cfg4_shard_n$n" --gpus $n --workload cfg4shard --steps 10 --warmup 3 --cpu-frames 0 --no-e2e
    run "cfg5_clips_n$n" --gpus $n --mode clips --steps 20 --warmup 5 --cpu-frames 0 --no-e2e
    run "e2e_n$n" --gpus $n --mode e2e --steps 5 --warmup 1
done
if [ "$DRY" = 1 ]; then echo "python tools/capi_shard_run.py --gpus $MAX"; exit 0; fi
echo "== capi: python tools/capi_shard_run.py --gpus $MAX" | tee -a "$OUT/scale.jsonl" >&2
timeout 900 python tools/capi_shard_run.py --gpus "$MAX" >> "$OUT/scale.jsonl" 2> "$OUT/capi.err" || echo "{\"failed\": \"capi\", \"rc\": $?}" >> "$OUT/scale.jsonl"
grep -c '"value"' "$OUT/scale.jsonl" >&2
