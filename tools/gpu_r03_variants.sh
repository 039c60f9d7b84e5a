#!/bin/bash
# GPU box: rocprofv3 kernel stats of `bench.py --workload W` for every library build_variants/lib_*.so (lab variants), and the shipped one
set -u
TAG=${1:-r03v}; W=${2:-C5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
cp cerbos_amd/libcerbos_hip.so /tmp/lib_orig.so
for lib in /tmp/lib_orig.so build_variants/lib_v_*.so; do
  [ -f "$lib" ] || continue
  name=$(basename $lib .so)
  cp $lib cerbos_amd/libcerbos_hip.so
  BENCH="python $R/bench.py --workload $W --batches 8 --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs"
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o r -- $BENCH > $OUT/prof_$name.log 2>&1 )
  DB=$(find $OUT/prof_$name -name '*.db' | head -1)
  echo "== $name"
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_$name.txt | sed -n 3,5p
  rm -rf $OUT/prof_$name
done
cp /tmp/lib_orig.so cerbos_amd/libcerbos_hip.so
