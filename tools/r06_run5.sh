O=gpurun_out/r06e; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q > $O/bench_contract.txt 2>&1; tail -4 $O/bench_contract.txt
timeout 1500 python tools/fuzz_warp.py 8000 6101 > $O/r6_fz1.log 2>&1; tail -2 $O/r6_fz1.log
timeout 1500 python tools/fuzz_warp.py 1000 6102 large > $O/r6_fz2.log 2>&1; tail -1 $O/r6_fz2.log
timeout 1500 python tests/fuzz_parity.py 5000 6103 > $O/r6_fz3.log 2>&1; tail -1 $O/r6_fz3.log
timeout 1500 python tests/fuzz_parity.py 200 6104 -1 big > $O/r6_fz4.log 2>&1; tail -1 $O/r6_fz4.log
timeout 1500 python tools/fuzz_warp.py 3000 6105 tiny > $O/r6_fz5.log 2>&1; tail -1 $O/r6_fz5.log
