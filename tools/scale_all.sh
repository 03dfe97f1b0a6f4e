#!/bin/bash
# The whole north-star table in ONE command, for the first box that has more than one GPU (nothing here has ever run on more than one):
#
#     tools/scale_all.sh [MAXGPUS] [OUTDIR]          default: every GPU of the node, gpurun_out/scale
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
# (kept on purpose: the host driver of this pool only supports dmabuf IPC -- without it RCCL and cross-process tensor sharing fail with
# "hipIpcGetMemHandle: invalid argument"; see the Environment notes of the build)
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NGPU=$(python3 -c "import bench; print(bench.visible_gpus() or 1)")
MAX=${1:-$NGPU}
OUT=${2:-gpurun_out/scale}
mkdir -p "$OUT"
: > "$OUT/scale.jsonl"
run() {     # name, then the bench flags
    local name=$1; shift
    echo "== $name: python bench.py $*" | tee -a "$OUT/scale.jsonl" >&2
    if timeout 1500 python bench.py "$@" >> "$OUT/scale.jsonl" 2> "$OUT/$name.err"; then :; else
        echo "{\"failed\": \"$name\", \"rc\": $?}" >> "$OUT/scale.jsonl"
    fi
}
for n in 1 2 4 8; do
    [ "$n" -le "$MAX" ] || continue
    run "cfg2_shard_n$n" --gpus $n --steps 20 --warmup 5 --cpu-frames 0 --no-e2e
done
for n in 2 4 8; do
    [ "$n" -le "$MAX" ] || continue
    run "cfg4_shard_n$n" --gpus $n --workload cfg4shard --steps 10 --warmup 3 --cpu-frames 0 --no-e2e
    run "cfg5_clips_n$n" --gpus $n --mode clips --steps 20 --warmup 5 --cpu-frames 0 --no-e2e
    run "e2e_n$n" --gpus $n --mode e2e --steps 5 --warmup 1
done
echo "== capi: python tools/capi_shard_run.py --gpus $MAX" | tee -a "$OUT/scale.jsonl" >&2
timeout 900 python tools/capi_shard_run.py --gpus "$MAX" >> "$OUT/scale.jsonl" 2> "$OUT/capi.err" || echo "{\"failed\": \"capi\", \"rc\": $?}" >> "$OUT/scale.jsonl"
grep -c '"value"' "$OUT/scale.jsonl" >&2
