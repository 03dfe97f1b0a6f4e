#!/bin/bash
# Timing-only variants of the footprint-plan kernel (tools/time_plan.py times them; their plans are not fit for the warp):
#   exp1 launch + staging + range tables   exp2 + candidate shortlist   exp3 + classification and corner mapping   (product: + certificates, window)   exp4 the product with at most one trip of the candidate loop
set -e
cd "$(dirname "$0")/../meshflow_amd/csrc"
make -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
mkdir -p ../variants
build() {   # name, defines
    mkdir -p ../build/var_$1
    /opt/rocm/bin/hipcc $FLAGS $2 -c cell_table.hip -o ../build/var_$1/cell_table.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared ../build/var_$1/cell_table.o $(ls ../build/*.o | grep -v "/cell_table.o") -o ../variants/libmf_$1.so -ldl -lpthread
}
for m in 1 2 3 4; do build planexp$m -DMF_PLAN_EXP=$m & done
wait
ls ../variants/ | grep planexp
