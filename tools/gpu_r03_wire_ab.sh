#!/bin/bash
# GPU box: the wire kernels with nothing staged in LDS (CBH_WIRE_LDS=0), the assembler's outputs staged (1), the flattener's
# messages too (2) - each mode verified byte for byte against the host road before it is timed; short hard timeouts.
set -u
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r03w5}; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
CBH_WIRE_LDS=0 timeout -s KILL 240 python -m pytest tests/test_gpu_wire.py tests/test_per_call_globals.py -m gpu -x -q > $O/pytest_wire.log 2>&1; echo "pytest (mode 0) rc $?"; tail -2 $O/pytest_wire.log
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$PWD/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
python tools/export_wire.py C2 524288 /tmp/wire_C2 > $O/export_C2.log 2>&1
python tools/export_wire.py C5 262144 /tmp/wire_C5 > $O/export_C5.log 2>&1
for mode in 0 1 2; do
  for w in C2 C5; do
    echo "== mode $mode $w"
    CBH_WIRE_LDS=$mode timeout -s KILL 60 /tmp/e2e_wire_bench /tmp/wire_$w 65536 0.5 1 device_out verify > $O/verify_${w}_$mode.json 2>$O/verify_${w}_$mode.err
    rc=$?; tail -1 $O/verify_${w}_$mode.json; [ $rc -ne 0 ] && { echo "mode $mode $w FAILED rc $rc"; tail -2 $O/verify_${w}_$mode.err; continue; }
    CBH_WIRE_LDS=$mode timeout -s KILL 60 /tmp/e2e_wire_bench /tmp/wire_$w 131072 2 1,4 device_out | tee $O/e2e_${w}_$mode.json | grep road
  done
  ( cd /tmp && CBH_WIRE_LDS=$mode timeout -s KILL 60 rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o r -- /tmp/e2e_wire_bench /tmp/wire_C2 131072 1 1 device_out > $O/prof_$mode.log 2>&1 )
  DB=$(find $O/prof_$mode -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $O/kernel_stats_wire_C2_mode$mode.txt | sed -n 3,9p
  rm -rf $O/prof_$mode
done
