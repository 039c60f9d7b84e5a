#!/bin/bash
# GPU box: the walk's prologue and principal pass in parts.  The profiling library is built beforehand, in the build container:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCBH_PROFILE_CYCLES=2 -Iinclude cerbos_amd/csrc/cbh_engine.hip -o build_variants/lib_profile2.so
# (build_variants/ is git-ignored and travels with gpurun; -DCBH_PROFILE_CYCLES without a value gives lib_profile.so for the r03 scripts)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-r04_parts}; mkdir -p $OUT; cd $R
cp cerbos_amd/libcerbos_hip.so /tmp/lib_orig.so
cp build_variants/lib_profile2.so cerbos_amd/libcerbos_hip.so
for w in ${2:-C5}; do
  CBH_PROFILE_PARTS=1 timeout 200 python tools/gpu_cycles_walk2.py $w > $OUT/cycles_parts_$w.txt 2>&1
  cat $OUT/cycles_parts_$w.txt
done
cp /tmp/lib_orig.so cerbos_amd/libcerbos_hip.so
