#!/bin/bash
# Round 5: messages to a wave of the wire kernels (WireArgs.mpw / WireOutArgs.mpw: 64, 32, 16) - the kernels' own times, the roads'
# rates, the outputs verified against the host road at every setting.  Every step bounded; the first failure ends the script.
#   usage: gpu_r05_mpw.sh TAG [messages]   (the knobs exist only in the experiment's build: profiles/r05_wire_mpw_ab.txt)
set -u
TAG=${1:-r05m}; NMSG=${2:-262144}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$R/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
for w in C2 C5; do
  timeout 300 python tools/export_wire.py $w $NMSG /tmp/wire_$w > $OUT/export_$w.log 2>&1 || { echo "export $w failed"; exit 1; }
  for m in 64 32 16; do
    export CBH_WIRE_MPW=$m CBH_WIRE_OUT_MPW=$m
    timeout -k 3 40 /tmp/e2e_wire_bench /tmp/wire_$w 65536 0.5 1 device_out verify > $OUT/e2e_${w}_mpw$m.json 2>$OUT/e2e_${w}_mpw$m.err || { echo "$w mpw $m FAILED"; tail -3 $OUT/e2e_${w}_mpw$m.err; exit 1; }
    grep -h "verified\|road" $OUT/e2e_${w}_mpw$m.json | tr '\n' ' '; echo
    ( cd /tmp && timeout -k 3 60 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w$m -o r -- /tmp/e2e_wire_bench /tmp/wire_$w 131072 2 1 device_out > $OUT/prof_${w}_mpw$m.log 2>&1 ) || { echo "$w mpw $m rocprof FAILED"; tail -3 $OUT/prof_${w}_mpw$m.log; exit 1; }
    DB=$(find $OUT/prof_$w$m -name '*.db' | head -1)
    [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_wire_${w}_mpw$m.txt | grep "cbh_wire_\(fill\|count\|out_write\|out_size\|scan\)" | head -6 | cut -c1-120
    rm -rf $OUT/prof_$w$m
    timeout -k 3 40 /tmp/e2e_wire_bench /tmp/wire_$w $NMSG 3 1 onecall 2>&1 | grep road | sed "s/^/$w mpw $m /"
  done
  unset CBH_WIRE_MPW CBH_WIRE_OUT_MPW
  echo "$w default:"; timeout -k 3 40 /tmp/e2e_wire_bench /tmp/wire_$w $NMSG 3 1 onecall 2>&1 | grep road
done 2>&1 | tee $OUT/mpw_ab.txt
