#!/bin/bash
# kernel-trace stats of bench.py (our kernels only)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o r -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-frames 0 > /tmp/ks.log 2>&1
grep -E "mf::" /tmp/ks/r_kernel_stats.csv | awk -F'","|",|,"' '{n=split($1,a,"("); printf "%-60s calls=%s avg_ns=%s\n", substr(a[1],2,58), $2, $4}'
