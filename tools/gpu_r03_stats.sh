#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of `bench.py --workload W` (no side legs) -> gpurun_out/<tag>/kernel_stats_W.txt
set -u
TAG=${1:-r03s}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for w in "$@"; do
  NB=$(case $w in C2) echo 12;; C3) echo 3;; C4) echo 6;; C5) echo 8;; *) echo 4;; esac)
  BENCH="python $R/bench.py --workload $w --batches $NB --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- $BENCH > $OUT/prof_$w.log 2>&1 )
  DB=$(find $OUT/prof_$w -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_$w.txt | head -8
  rm -rf $OUT/prof_$w
done
