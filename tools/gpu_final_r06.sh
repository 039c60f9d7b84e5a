#!/bin/bash
# Runs on the GPU box (via gpurun): the judged round-6 artefacts of the SHIPPED build (the in-tree .so files as they travelled).
# Every step bounded; everything lands in gpurun_out/<tag>/ - copy what should be judged into profiles/.
#   1. the GPU tier, smoke()                                                                         -> pytest_gpu.log, smoke.log
#   2. per workload: rocprofv3 --kernel-trace --stats of `bench.py --workload W` (no side legs)      -> kernel_stats_W.txt
#      and rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one pass each                     -> pmc_W/
#   3. counter calibration on known byte counts (tools/pmc_calib.hip); tools/pmc_summary.py          -> pmc_traffic.json,
#      copied to profiles/r06_pmc_traffic.json ON THE BOX so that the bench lines below carry this call's traffic
#   4. python bench.py (the driver's flags) and the other workloads' lines                           -> bench_W.json
set -u
TAG=${1:-r06f}
WL=${2:-"C2 C3 C4 C5 T"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
(time timeout -k 5 900 python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
timeout -k 5 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
for w in $WL; do
  NB=$(case $w in C2) echo 12;; C3) echo 3;; C4) echo 6;; *) echo 8;; esac)
  BENCH="python $R/bench.py --workload $w --batches $NB --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs"
  ( cd /tmp && CBH_BENCH_NO_DIST=1 timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- $BENCH > $OUT/prof_$w.log 2>&1 )
  DB=$(find $OUT/prof_$w -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_$w.txt | sed -n 3,7p | cut -c1-130
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && CBH_BENCH_NO_DIST=1 timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$w/$c -o $c -- $BENCH > $OUT/pmc_${w}_$c.log 2>&1 )
  done
  rm -rf $OUT/prof_$w
done
hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib > $OUT/calib_build.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_calib/$c -o $c -- /tmp/pmc_calib > $OUT/pmc_calib_$c.log 2>&1 )
done
python tools/pmc_summary.py $OUT | tail -1 | cut -c1-600
[ -s $OUT/pmc_traffic.json ] && cp $OUT/pmc_traffic.json profiles/r06_pmc_traffic.json
(timeout -k 5 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench_C2.err | grep '^{' | tail -1) > $OUT/bench_C2.json
cut -c1-330 $OUT/bench_C2.json; echo
for w in C3 C4 C5 T C5W; do
  (timeout -k 5 400 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>$OUT/bench_$w.err | grep '^{' | tail -1) > $OUT/bench_$w.json
  python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']; s=r.get('serial') or {}
print('$w', '%.4g dec/s' % d['value'], r['kernel'], '%.1f us' % (r['kernel_ms']*1e3), 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3), 'traffic', r['traffic'])"
done
find $OUT -name '*.csv' -size +300k -delete
