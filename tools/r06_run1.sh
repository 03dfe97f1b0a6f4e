mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
(cd tools && timeout 300 ./ubench_lds_rowmap) > $O/ubench_rowmap.txt 2>&1
timeout 900 python tools/ab_warp.py --workloads cfg2,cfg3,cfg4shard --rounds 5 meshflow_amd/libmeshflow_hip.so meshflow_amd/variants/libmf_rowmap.so > $O/ab_rowmap.txt 2>&1
timeout 600 python tools/phase_profile.py cfg3 > $O/phase_cfg3.txt 2>&1
SETS=2 timeout 900 bash tools/pmc_warp.sh meshflow_amd/libmeshflow_hip.so cfg3 r06base_cfg3 > /dev/null 2>&1
timeout 900 python bench.py > $O/bench_base.json 2> $O/bench_base.err
tail -5 $O/ubench_rowmap.txt; cat $O/ab_rowmap.txt | tail -12
