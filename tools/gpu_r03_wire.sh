#!/bin/bash
# GPU box: the device flattener + assembler (cbh_wire.h) - GPU tests, then the wire-inclusive end-to-end rate by the three roads
# (tools/e2e_wire_bench.cpp: serialized CheckInputs in, serialized CheckOutputs out) and the kernels' own times (rocprofv3).
# usage: gpu_r03_wire.sh TAG [all|wire|notests] [workloads]
set -u
TAG=${1:-r03w}; TESTS=${2:-wire}; WL=${3:-"C2 C5"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "$TESTS" = all ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_gpu.log
elif [ "$TESTS" = wire ]; then
  timeout 600 python -m pytest tests/test_gpu_wire.py -x -q > $OUT/pytest_wire.log 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest_wire.log
fi
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$R/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
for w in $WL; do
  python tools/export_wire.py $w 524288 /tmp/wire_$w > $OUT/export_$w.log 2>&1
  timeout 300 /tmp/e2e_wire_bench /tmp/wire_$w 65536 1 1 both verify > $OUT/e2e_${w}_verify.json 2>$OUT/e2e_${w}_verify.err; tail -4 $OUT/e2e_${w}_verify.json; tail -3 $OUT/e2e_${w}_verify.err
  for S in 16384 131072; do
    timeout 300 /tmp/e2e_wire_bench /tmp/wire_$w $S 2 1,4,16,32 both > $OUT/e2e_${w}_$S.json 2>$OUT/e2e_${w}_$S.err; cat $OUT/e2e_${w}_$S.json; tail -3 $OUT/e2e_${w}_$S.err
  done
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- /tmp/e2e_wire_bench /tmp/wire_$w 131072 2 1 device_out > $OUT/prof_$w.log 2>&1 )
  DB=$(find $OUT/prof_$w -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_wire_$w.txt | head -14
  rm -rf $OUT/prof_$w
done
