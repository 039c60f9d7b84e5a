#!/bin/bash
# Round 5: the wire kernels after the move to typed pointers / 8-byte windows (cbh_wire.h): parity on the hardware first, then the
# kernels' own times per 131072 messages, staged in LDS against parsed in place, then the road's rate from one caller thread.
#   usage: gpu_r05_wire.sh TAG [notests]
set -u
TAG=${1:-r05w}; TESTS=${2:-tests}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "$TESTS" = tests ]; then
  timeout 900 python -m pytest tests/test_gpu_wire.py tests/test_zz_gpu_request_road.py -m gpu -x -q > $OUT/pytest_wire.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_wire.log
fi
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$R/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
for w in C2 C5; do
  python tools/export_wire.py $w 262144 /tmp/wire_$w > $OUT/export_$w.log 2>&1
  timeout 300 /tmp/e2e_wire_bench /tmp/wire_$w 65536 1 1 both verify > $OUT/e2e_${w}_verify.json 2>$OUT/e2e_${w}_verify.err; tail -2 $OUT/e2e_${w}_verify.json; tail -2 $OUT/e2e_${w}_verify.err
  for cap in default 0; do
    [ $cap = default ] && unset CBH_WIRE_FILL_LDS_MAX || export CBH_WIRE_FILL_LDS_MAX=$cap
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- /tmp/e2e_wire_bench /tmp/wire_$w 131072 2 1 device_out > $OUT/prof_${w}_$cap.log 2>&1 )
    DB=$(find $OUT/prof_$w -name '*.db' | head -1)
    echo "== $w fill staging: $cap"
    [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_wire_${w}_lds_$cap.txt | head -17
    rm -rf $OUT/prof_$w
  done
  unset CBH_WIRE_FILL_LDS_MAX
  timeout 300 python tools/gpu_requests_and_trail.py $w 250000 10 > $OUT/requests_and_trail_$w.txt 2>&1; grep "decisions/s" $OUT/requests_and_trail_$w.txt | head -3
done
