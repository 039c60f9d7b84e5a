#!/bin/bash
# The FIRST GPU call of the next round (the last build of round 3 ended with its GPU minutes spent): the whole GPU tier on
# the build with the walk's wider shapes, then the A/B nobody has timed yet - C5W (C5's table, principals with five to eight
# roles) through cbh_walk2_wide_kernel against CBH_NO_WALK2_WIDE=1 (the general walk, as before) - with rocprofv3 kernel
# statistics of both, and a bench line per workload to compare with profiles/r03_bench_*.json.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r04_first.sh r04a'
set -u
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
(time timeout -s KILL 600 python -m pytest tests -m gpu -x -q) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
for mode in wide general; do
  [ $mode = general ] && export CBH_NO_WALK2_WIDE=1 || unset CBH_NO_WALK2_WIDE
  BENCH="python $R/bench.py --workload C5W --batches 8 --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_C5W_$mode -o r -- $BENCH > $OUT/prof_C5W_$mode.log 2>&1 )
  DB=$(find $OUT/prof_C5W_$mode -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_C5W_$mode.txt | sed -n 3,8p
  rm -rf $OUT/prof_C5W_$mode
  (timeout 600 python bench.py --workload C5W --steps 10 --warmup 2 --no-cpu-baseline 2>$OUT/bench_C5W_$mode.err | grep '^{' | tail -1) > $OUT/bench_C5W_$mode.json
  python - <<P
import json
try:
    d = json.load(open('$OUT/bench_C5W_$mode.json')); r = d['roofline']; s = r.get('serial') or {}
    print('C5W $mode', '%.3g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3))
except Exception as e:
    print('C5W $mode failed', e); print(open('$OUT/bench_C5W_$mode.err').read()[-1500:])
P
done
unset CBH_NO_WALK2_WIDE
bash tools/gpu_quick_r03.sh $TAG "C2 C3 C4 C5 T" notests
