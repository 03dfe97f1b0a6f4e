O=gpurun_out/r06f; mkdir -p $O
(cd tools && timeout 300 ./ubench_lds_u16) > $O/ubench_u16.txt 2>&1; cat $O/ubench_u16.txt
timeout 1500 python tools/ab_warp.py --workloads cfg2,cfg3,cfg4shard --rounds 7 meshflow_amd/libmeshflow_hip.so meshflow_amd/variants/libmf_u16.so > $O/ab_u16.txt 2>&1; grep -v "^$" $O/ab_u16.txt | tail -9
timeout 600 python tools/class_census.py cfg2 cfg3 > $O/census.txt 2>&1; grep -i "multi" $O/census.txt
timeout 1500 python tools/phase_variant_check.py cfg3 > $O/variant_check.txt 2>&1; cat $O/variant_check.txt
