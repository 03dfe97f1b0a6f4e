# A/B two builds of libmeshflow_hip.so on the same GPU box: put the other build at meshflow_amd/libmeshflow_hip_prev.so, then
#   gpurun -- bash tools/ab_libs.sh
cd $GRAFT_REPO_ROOT
cp meshflow_amd/libmeshflow_hip.so /tmp/new.so
run() { python bench.py --steps 30 --warmup 5 --cpu-frames 0 --workload $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', '$1', round(d['ms_per_step'],3), d['roofline']['avg_launch_ms'])"; }
for rep in 1 2 3; do
  cp /tmp/new.so meshflow_amd/libmeshflow_hip.so; run cfg2 new; run cfg3 new
  cp meshflow_amd/libmeshflow_hip_prev.so meshflow_amd/libmeshflow_hip.so; run cfg2 prev; run cfg3 prev
done
