O=gpurun_out/r06d; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/gputests.txt 2>&1
tail -12 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
