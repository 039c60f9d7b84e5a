#!/bin/bash
# Round 5: two builds of the library (ab_old.so / ab_new.so at the repo root) on ONE box - the wire kernels' own times on C5 and C2,
# the outputs verified against the host road on the new build.   usage: gpu_r05_wire_ab.sh TAG
set -u
TAG=${1:-r05wab}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$R/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
for w in C5 C2; do
  timeout 300 python tools/export_wire.py $w 131072 /tmp/wire_$w > $OUT/export_$w.log 2>&1 || { echo "export $w failed"; exit 1; }
  for v in old new old new; do
    cp ab_$v.so cerbos_amd/libcerbos_hip.so
    ( cd /tmp && timeout -k 3 60 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- /tmp/e2e_wire_bench /tmp/wire_$w 131072 1.5 1 device_out > $OUT/prof.log 2>&1 ) || { echo "$w $v rocprof FAILED"; tail -3 $OUT/prof.log; }
    DB=$(find $OUT/prof -name '*.db' | head -1)
    [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" /dev/null | grep "cbh_wire_fill" | head -1 | cut -c1-120 | sed "s/^/$w $v /"
    rm -rf $OUT/prof
  done
  cp ab_new.so cerbos_amd/libcerbos_hip.so
  timeout -k 3 60 /tmp/e2e_wire_bench /tmp/wire_$w 65536 0.3 1 device_out verify 2>&1 | grep -h "verified\|FAILED" | sed "s/^/$w new /"
done 2>&1 | tee $OUT/ab.txt
