O=gpurun_out/r06b; mkdir -p $O
timeout 300 python tools/class_census.py cfg2 cfg3 cfg4shard > $O/census.txt 2>&1
timeout 1200 python tools/ab_warp.py --workloads cfg2,cfg3,cfg4shard --rounds 5 meshflow_amd/variants/libmf_r05.so meshflow_amd/variants/libmf_rowmap.so meshflow_amd/libmeshflow_hip.so > $O/ab.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1
cat $O/census.txt; grep -v "^$" $O/ab.txt | tail -14; tail -5 $O/gputests.txt
