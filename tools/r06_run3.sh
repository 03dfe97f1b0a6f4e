O=gpurun_out/r06c; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1
tail -6 $O/gputests.txt
timeout 1500 python tools/ab_warp.py --workloads cfg2,cfg3,cfg4shard --rounds 9 meshflow_amd/variants/libmf_r05.so meshflow_amd/libmeshflow_hip.so > $O/ab.txt 2>&1
grep -v "^$" $O/ab.txt | tail -9
timeout 2400 bash tools/profile_r05.sh r06a > $O/profile.log 2>&1
tail -30 $O/profile.log
