#!/bin/bash
# Builds the experiment variants of the warp kernel that tools/phase_profile.py times (hipcc cross-compiles: no GPU needed).
set -e
cd "$(dirname "$0")/../meshflow_amd/csrc"
make -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
mkdir -p ../variants
build() {   # name, define
    mkdir -p ../build/var_$1
    /opt/rocm/bin/hipcc $FLAGS $2 -c warp.hip -o ../build/var_$1/warp.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared ../build/var_$1/warp.o $(ls ../build/*.o | grep -v "/warp.o") -o ../variants/libmf_$1.so -ldl -lpthread
}
for m in 1 2 4 8 16 31 32 64; do build skip$m -DMF_EXP_SKIP=$m & done
build phases -DMF_EXP_PHASES=1 &
wait
ls ../variants/
# a timing-only build may compile away the code that consumed an asynchronous scalar load: check every variant on the CPU before it goes
# to a GPU (round 6: three of these builds faulted on the GPU for exactly that reason; tools/isa_guard.py saw it in the disassembly)
cd ../.. && for v in meshflow_amd/variants/libmf_skip*.so meshflow_amd/variants/libmf_phases.so; do python tools/isa_guard.py $v --hazards-only > /dev/null || { echo "HAZARD in $v"; python tools/isa_guard.py $v --hazards-only; exit 1; }; done; echo "all variants: no in-flight scalar-load hazard"
