O=gpurun_out/r06h; mkdir -p $O
timeout 1800 python tools/ab_warp.py --workloads cfg2,cfg3,cfg4shard --rounds 7 meshflow_amd/libmeshflow_hip.so meshflow_amd/variants/libmf_pf1.so meshflow_amd/variants/libmf_pf2.so > $O/ab_pf.txt 2>&1; grep -v "^$" $O/ab_pf.txt | tail -15
