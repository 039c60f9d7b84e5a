#!/bin/bash
# GPU box: per-wave phase profile of the walk2 kernels (profiling library build_variants/lib_profile.so)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-r03_cycles_w2}
mkdir -p $OUT
cd $R
cp cerbos_amd/libcerbos_hip.so /tmp/lib_orig.so
cp build_variants/lib_profile.so cerbos_amd/libcerbos_hip.so
for w in ${2:-C5}; do
  timeout 200 python tools/gpu_cycles_walk2.py $w > $OUT/cycles_walk_$w.txt 2>&1
  CBH_PRE_ONLY=1 timeout 200 python tools/gpu_cycles_walk2.py $w > $OUT/cycles_pre_$w.txt 2>&1
  cat $OUT/cycles_walk_$w.txt $OUT/cycles_pre_$w.txt
done
cp /tmp/lib_orig.so cerbos_amd/libcerbos_hip.so
