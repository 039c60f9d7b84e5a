#!/bin/bash
# Round 5: wire kernels' own times + the one-call road's rate and phase marks, C2 and C5.   usage: gpu_r05_wire2.sh TAG
set -u
TAG=${1:-r05w3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_wire.py tests/test_zz_gpu_request_road.py -m gpu -x -q > $OUT/pytest_wire.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest_wire.log
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$R/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
for w in C2 C5; do
  python tools/export_wire.py $w 262144 /tmp/wire_$w > $OUT/export_$w.log 2>&1
  timeout 300 /tmp/e2e_wire_bench /tmp/wire_$w 65536 1 1 both verify > $OUT/e2e_${w}_verify.json 2>$OUT/e2e_${w}_verify.err; tail -1 $OUT/e2e_${w}_verify.err
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- /tmp/e2e_wire_bench /tmp/wire_$w 131072 2 1 device_out > $OUT/prof_${w}.log 2>&1 )
  DB=$(find $OUT/prof_$w -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_wire_${w}.txt | head -18
  rm -rf $OUT/prof_$w
  CBH_TRACE=1 python tools/gpu_wire_onecall.py $w 250000 2>&1 | tail -5 | cut -c1-330
  for s in 3 5 6; do CBH_WIRE_SLICES=$s python tools/gpu_wire_onecall.py $w 250000 2>&1 | tail -1; done
  CBH_WIRE_LAST_SLICE=100 python tools/gpu_wire_onecall.py $w 250000 2>&1 | tail -1 | sed 's/^/even slices: /'
  timeout 300 python tools/gpu_requests_and_trail.py $w 250000 10 > $OUT/requests_and_trail_$w.txt 2>&1; grep "decisions/s" $OUT/requests_and_trail_$w.txt | head -3
done
