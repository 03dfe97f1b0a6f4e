O=gpurun_out/r06g; mkdir -p $O
timeout 1500 python tools/phase_variant_check.py cfg3 > $O/variant_check.txt 2>&1; cat $O/variant_check.txt
if grep -q "rc=  -6" $O/variant_check.txt; then echo "a variant still faults: phase profile not run"; exit 0; fi
timeout 1500 python tools/phase_profile.py cfg2 cfg3 > $O/phase_profile.txt 2>&1; cat $O/phase_profile.txt | cut -c1-200
