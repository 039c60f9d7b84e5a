#!/bin/bash
# GPU box: the whole GPU tier, the default bench line, and where the device road's host time goes (rocprofv3 HIP API stats).
set -u
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r03c}; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$PWD/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
python tools/export_wire.py C2 524288 /tmp/wire_C2 > $O/export.log 2>&1
for T in 1 4; do
  ( cd /tmp && timeout 200 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/hip_$T -o r -- /tmp/e2e_wire_bench /tmp/wire_C2 131072 1.5 $T device_out > $O/hip_$T.log 2>&1 )
  tail -2 $O/hip_$T.log
  F=$(find $O/hip_$T -name '*hip_api_stats.csv' | head -1); [ -n "$F" ] && { head -14 "$F" | cut -c1-160; cp "$F" $O/hip_api_stats_$T.csv; }
  rm -rf $O/hip_$T
done
