#!/bin/bash
# GPU box: per-wave phase profile of the flat kernel (profiling library build_variants/lib_profile.so)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-r03_cycles_flat}
mkdir -p $OUT
cd $R
cp cerbos_amd/libcerbos_hip.so /tmp/lib_orig.so
cp build_variants/lib_profile.so cerbos_amd/libcerbos_hip.so
for w in ${2:-T}; do
  CBH_RESIDENT_STREAMS=1 timeout 300 python tools/gpu_cycles_flat.py $w > $OUT/cycles_flat_$w.txt 2>&1
  cat $OUT/cycles_flat_$w.txt
done
cp /tmp/lib_orig.so cerbos_amd/libcerbos_hip.so
