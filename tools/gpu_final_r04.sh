#!/bin/bash
# Runs on the GPU box (via gpurun): the judged round-4 artefacts, in an order that lets the bench lines carry the PMC
# traffic measured in THIS call.
#   1. per workload: rocprofv3 --kernel-trace --stats of `bench.py --workload W` (no side legs)  -> kernel_stats_W.txt
#      and rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one pass each               -> pmc_W/
#   2. counter calibration on known byte counts (tools/pmc_calib.hip)                          -> pmc_calibration.json
#   3. tools/pmc_summary.py -> pmc_traffic.json, copied to profiles/r04_pmc_traffic.json ON THE BOX (what bench.py reads)
#   4. python bench.py (driver flags: --steps 20 --warmup 3, CPU baseline + side legs)         -> bench_C2.json
#      python bench.py --workload C3|C4|C5|T --steps 10 --warmup 2                              -> bench_W.json
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r04f}
WL=${2:-"C2 C3 C4 C5 T"}   # (C5W: bench line only, below)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
(time timeout -s KILL 900 python -m pytest tests -m gpu -x -q) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
for w in $WL; do
  NB=$(case $w in C2) echo 12;; C3) echo 3;; C4) echo 6;; C5) echo 8;; *) echo 8;; esac)
  BENCH="python $R/bench.py --workload $w --batches $NB --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o r -- $BENCH > $OUT/prof_$w.log 2>&1 )
  DB=$(find $OUT/prof_$w -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_$w.txt | sed -n 3,5p
  [ -z "${SKIP_PMC:-}" ] && for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$w/$c -o $c -- $BENCH > $OUT/pmc_${w}_$c.log 2>&1 )
  done
  rm -rf $OUT/prof_$w
done
if [ -z "${SKIP_PMC:-}" ]; then   # SKIP_PMC=1: the bench lines carry the traffic of the committed profiles/r04_pmc_traffic.json (an earlier call)
hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib > $OUT/calib_build.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_calib/$c -o $c -- /tmp/pmc_calib > $OUT/pmc_calib_$c.log 2>&1 )
done
python tools/pmc_summary.py $OUT > /dev/null
cp $OUT/pmc_traffic.json profiles/r04_pmc_traffic.json
fi
(timeout 900 python bench.py --steps 20 --warmup 3 2>$OUT/bench_C2.err | grep '^{' | tail -1) > $OUT/bench_C2.json
cut -c1-330 $OUT/bench_C2.json; echo
for w in $WL; do
  [ $w = C2 ] && continue
  (timeout 600 python bench.py --workload $w --steps 10 --warmup 2 2>$OUT/bench_$w.err | grep '^{' | tail -1) > $OUT/bench_$w.json
  python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']; print('$w', '%.3g dec/s' % d['value'], r['kernel'], '%.1f us' % (r['kernel_ms']*1e3), 'frac %.3f' % r['frac'], 'traffic', r['traffic'])"
done
(timeout 600 python bench.py --workload C5W --steps 10 --warmup 2 --no-cpu-baseline 2>$OUT/bench_C5W.err | grep '^{' | tail -1) > $OUT/bench_C5W.json
find $OUT -name '*.csv' -size +300k -delete
