#!/bin/bash
# GPU box: requests grouped by route on the device (cbh_wire_route_kernel ...) - the GPU tests by both roads, then the device
# road with (default) and without (CBH_WIRE_GROUP=0) the grouping on C5 and C3 streams in arrival order.
set -u
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r03g}; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -s KILL 300 python -m pytest tests/test_gpu_wire.py tests/test_per_call_globals.py tests/test_gpu_golden.py -m gpu -x -q > $O/pytest_wire.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_wire.log
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$PWD/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
for w in C5 C3; do
  python tools/export_wire.py $w 262144 /tmp/wire_$w > $O/export_$w.log 2>&1
  for g in 1 0; do
    echo "== $w CBH_WIRE_GROUP=$g"
    CBH_WIRE_GROUP=$g timeout -s KILL 60 /tmp/e2e_wire_bench /tmp/wire_$w 65536 0.5 1 device_out verify > $O/verify_${w}_$g.json 2>$O/verify_${w}_$g.err || { echo "FAILED"; tail -2 $O/verify_${w}_$g.err; continue; }
    tail -1 $O/verify_${w}_$g.json
    CBH_WIRE_GROUP=$g timeout -s KILL 60 /tmp/e2e_wire_bench /tmp/wire_$w 131072 2 1,4 device_out | tee $O/e2e_${w}_group$g.json | grep road
  done
done
( cd /tmp && timeout -s KILL 60 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- /tmp/e2e_wire_bench /tmp/wire_C5 131072 1 1 device_out > $O/prof.log 2>&1 )
DB=$(find $O/prof -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $O/kernel_stats_wire_C5_grouped.txt | sed -n 3,16p
rm -rf $O/prof
