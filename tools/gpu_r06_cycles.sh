#!/bin/bash
# Round 6: per-wave phase cycles of the flat kernels (a -DCBH_PROFILE_CYCLES build of the library at ab_prof.so; tools/gpu_cycles_flat.py),
# then the GPU tier on the shipped build.   usage: gpu_r06_cycles.sh TAG "C2 C3 T" [pytest args]
set -u
TAG=${1:-r06cyc}; WL=${2:-"C2 T"}; PT=${3:-""}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp CBH_BENCH_NO_DIST=1
cp cerbos_amd/libcerbos_hip.so /tmp/shipped.so
cp ab_prof.so cerbos_amd/libcerbos_hip.so
for w in $WL; do timeout -k 5 300 python tools/gpu_cycles_flat.py $w > $OUT/cycles_flat_$w.txt 2>$OUT/cycles_flat_$w.err; head -12 $OUT/cycles_flat_$w.txt; done
cp /tmp/shipped.so cerbos_amd/libcerbos_hip.so
if [ -n "$PT" ]; then (time timeout -k 5 900 python -m pytest $PT -m gpu -q) > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log; fi
