run() { echo -n "$1 :: "; shift; env "$@" | head -1; }
for kb in 2048 4096 6075 8192 12150 16384 24300 32768 40000 51200; do run "6+6 piece up $kb KB" FRAME_KB=$kb tools/pcie_staged 1780 6 6; done
for kb in 6075 16384 32768; do run "6+6 piece down $kb KB" FRAME_DN_KB=$kb tools/pcie_staged 1780 6 6; done
run "6+6 chunk 32" CHUNK_MB=32 tools/pcie_staged 1780 6 6
run "6+6 chunk 40" CHUNK_MB=40 tools/pcie_staged 1780 6 6
run "8+8 chunk 25" CHUNK_MB=25 tools/pcie_staged 1780 8 8
run "12+4 frames" FRAME_KB=6075 tools/pcie_staged 1780 12 4
run "16+4 frames" FRAME_KB=6075 tools/pcie_staged 1780 16 4
run "up only 1 thread 50MB" tools/pcie_staged 1780 1 0
run "up only 1 thread frames" FRAME_KB=6075 tools/pcie_staged 1780 1 0
run "up only 4 thread frames" FRAME_KB=6075 tools/pcie_staged 1780 4 0
run "up only 4 thread 50MB" tools/pcie_staged 1780 4 0
run "up only 6 thread 50MB" tools/pcie_staged 1780 6 0
run "down only 4" tools/pcie_staged 1780 0 4
for v in 1 4 16 64; do run "frames 6+6 PINNED_MIN=$v" GPU_PINNED_MIN_XFER_SIZE=$v FRAME_KB=6075 tools/pcie_staged 1780 6 6; done
for v in 4 8 64; do run "frames 6+6 PINNED_XFER=$v" GPU_PINNED_XFER_SIZE=$v FRAME_KB=6075 tools/pcie_staged 1780 6 6; done
for v in 1 8 16 64; do run "frames 6+6 STAGING=$v" GPU_STAGING_BUFFER_SIZE=$v FRAME_KB=6075 tools/pcie_staged 1780 6 6; done
for v in 1 16 64; do run "50MB 6+6 STAGING=$v" GPU_STAGING_BUFFER_SIZE=$v tools/pcie_staged 1780 6 6; done
run "50MB 6+6 PINNED_MIN=1" GPU_PINNED_MIN_XFER_SIZE=1 tools/pcie_staged 1780 6 6
run "50MB 6+6 PINNED_MIN=1024" GPU_PINNED_MIN_XFER_SIZE=1024 tools/pcie_staged 1780 6 6
