#!/bin/bash
# Round 5, first GPU call: what round 4 wrote after its GPU minutes ended, measured - minus the GPU tier itself (the driver ran it at
# round end on this very build: GPUTEST_r04.json, 352 passed) and minus the bench lines that the round's last call retakes anyway.
#   1. the wire kernels' own times (baseline for this round's work on them)         -> kernel_stats_wire_{C2,C5}.txt
#   2. the pre-pass in two kernels against the fused one (CBH_PRE_SPLIT)            -> bench_{C5,C5W}_presplit{0,1}.json
#   3. the request road against the input road, with and without the trail          -> requests_and_trail_{C2,C5}.txt
#   4. what the effective-policy trail costs the resident kernels                   -> bench_{C2,T,C5}_trail.json
#   5. C3's HBM traffic now that its column cache holds the tags as bytes           -> pmc_traffic.json (calibration: one pass each)
set -u
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$R/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
for w in C2 C5; do
  python tools/export_wire.py $w 262144 /tmp/wire_$w > $OUT/export_$w.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- /tmp/e2e_wire_bench /tmp/wire_$w 131072 2 1 device_out > $OUT/prof_$w.log 2>&1 )
  DB=$(find $OUT/prof_$w -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_wire_$w.txt | head -16
  rm -rf $OUT/prof_$w
done
for mode in 0 1; do
  for w in C5 C5W; do
    CBH_PRE_SPLIT=$mode CBH_BENCH_NO_DIST=1 timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs > $OUT/bench_${w}_presplit$mode.json 2> $OUT/bench_${w}_presplit$mode.err
    python -c "
import json; d=json.loads([l for l in open('$OUT/bench_${w}_presplit$mode.json') if l.startswith('{')][-1]); r=d['roofline']; s=r.get('serial') or {}
print('$w CBH_PRE_SPLIT=$mode', '%.3g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3))"
  done
done
CBH_PRE_SPLIT=1 timeout 300 python -m pytest tests/test_gpu_synthetic.py -m gpu -x -q -k "c5 or C5" 2>&1 | tail -3
for w in C2 C5; do
  timeout 300 python tools/gpu_requests_and_trail.py $w 250000 10 > $OUT/requests_and_trail_$w.txt 2>&1; tail -8 $OUT/requests_and_trail_$w.txt
done
for w in C2 T C5; do
  CBH_BENCH_NO_DIST=1 timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs --audit-trail > $OUT/bench_${w}_trail.json 2> $OUT/bench_${w}_trail.err
  python -c "
import json; d=json.loads([l for l in open('$OUT/bench_${w}_trail.json') if l.startswith('{')][-1]); r=d['roofline']; s=r.get('serial') or {}
print('$w with the audit trail', '%.3g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3))"
done
BENCH="python $R/bench.py --workload C3 --batches 3 --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs"
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && CBH_BENCH_NO_DIST=1 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_C3/$c -o $c -- $BENCH > $OUT/pmc_C3_$c.log 2>&1 )
done
hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib > $OUT/calib_build.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_calib/$c -o $c -- /tmp/pmc_calib > $OUT/pmc_calib_$c.log 2>&1 )
done
python tools/pmc_summary.py $OUT | tail -1
find $OUT -name '*.csv' -size +300k -delete
