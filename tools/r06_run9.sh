O=gpurun_out/r06i; mkdir -p $O
timeout 1800 python tools/ab_warp.py --workloads cfg2,cfg3,cfg4shard --rounds 9 meshflow_amd/variants/libmf_head.so meshflow_amd/libmeshflow_hip.so > $O/ab_kernarg.txt 2>&1; grep -v "^$" $O/ab_kernarg.txt | tail -9
