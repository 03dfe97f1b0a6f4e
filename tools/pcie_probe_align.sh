# host address alignment (tools/pcie_staged, `direct` line only; chunks of 8 frames of 1080p)
run() { echo -n "$1 :: "; shift; env "$@" | head -1; }
for ud in "6 6" "4 4"; do
for rep in 1 2; do
run "$ud aligned" CHUNK_KB=48600 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud up+16" OFF_UP=16 CHUNK_KB=48600 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud dn+16" OFF_DN=16 CHUNK_KB=48600 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud both+16" OFF_UP=16 OFF_DN=16 CHUNK_KB=48600 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
run "$ud both+64" OFF_UP=64 OFF_DN=64 CHUNK_KB=48600 KERNEL=1 RING=18 tools/pcie_staged 1780 $ud
done
done
