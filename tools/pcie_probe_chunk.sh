# chunk size in whole 1080p frames (6,075 KB each) through the prototype with the pipeline's dependencies, and through the pipeline itself
run() { echo -n "$1 :: "; shift; env "$@" | head -1; }
for f in 8 9 10 11 12 16; do
  kb=$((f * 6075))
  run "proto 6+6 chunk $f frames, ring 18" CHUNK_KB=$kb KERNEL=1 RING=18 tools/pcie_staged 1780 6 6
  run "proto 4+4 chunk $f frames, ring 18" CHUNK_KB=$kb KERNEL=1 RING=18 tools/pcie_staged 1780 4 4
done
run "proto 6+6 50 MiB again" KERNEL=1 RING=18 tools/pcie_staged 1780 6 6
export GRID="6,6,8,18;6,6,9,18;6,6,10,16;6,6,11,16;6,6,12,14;6,6,16,10;6,4,9,18;6,4,11,16;4,4,9,18;4,4,11,16;4,4,8,18"
CONTIG=1 python tools/time_e2e_crop.py grid cfg2 2>&1
