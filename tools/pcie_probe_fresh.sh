# fresh output pages (np.empty of the caller) and transparent huge pages beside the duplex copies (tools/pcie_staged, `direct` line only)
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>&1
run() { echo -n "$1 :: "; shift; env "$@" | head -1; }
for ud in "6 6" "4 4"; do
run "$ud pipeline" KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud pipeline FRESH=8" FRESH=8 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud pipeline FRESH=4" FRESH=4 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud pipeline FRESH=8 THP out" THP=1 FRESH=8 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud pipeline FRESH=2 THP out" THP=1 FRESH=2 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud pipeline FRESH=8 THP both" THP=3 FRESH=8 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud pipeline THP both" THP=3 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud pipeline THP in, frames" THP=2 FRAME_KB=6075 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
done
