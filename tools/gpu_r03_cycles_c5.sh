#!/bin/bash
# GPU box: per-wave phase cycles + ablations of the general walk on C5 (profiling library build_variants/lib_profile.so)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r03_cycles
mkdir -p $OUT
cd $R
cp cerbos_amd/libcerbos_hip.so /tmp/lib_orig.so
cp build_variants/lib_profile.so cerbos_amd/libcerbos_hip.so
timeout 300 python tools/gpu_cycles.py C5 > $OUT/cycles_C5.txt 2>&1
cp /tmp/lib_orig.so cerbos_amd/libcerbos_hip.so
cat $OUT/cycles_C5.txt
