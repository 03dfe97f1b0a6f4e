# which of the pipeline's ingredients costs the duplex rate?  (tools/pcie_staged, `direct` line only)
run() { echo -n "$1 :: "; shift; env "$@" | head -1; }
for ud in "4 4" "6 6"; do
run "$ud plain" tools/pcie_staged 1780 $ud
run "$ud DEP" DEP=1 tools/pcie_staged 1780 $ud
run "$ud DEP+KERNEL" KERNEL=1 tools/pcie_staged 1780 $ud
run "$ud DEP+RING18" RING=18 tools/pcie_staged 1780 $ud
run "$ud DEP+KERNEL+RING18" KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud DEP+KERNEL+RING18 frames" FRAME_KB=6075 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
done
run "6 4 DEP+KERNEL+RING18" KERNEL=1 RING=18 tools/pcie_staged 1780 6 4
run "6 6 DEP+KERNEL+RING36" KERNEL=1 RING=36 tools/pcie_staged 1780 6 6
run "6 6 DEP+KERNEL+RING8" KERNEL=1 RING=8 tools/pcie_staged 1780 6 6
