#!/bin/bash
# The mask walk (cbh_check_flat_kernel_masks) on the GPU: parity on T / C4 (every tuple against oracle/ccheck.cpp), then bench lines
# and rocprofv3 kernel statistics with the mask walk on and off.   usage: gpu_r04_masks.sh TAG "T C4" [tests]
set -u
TAG=${1:-r04m}; WL=${2:-"T C4"}; TESTS=${3:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ -n "$TESTS" ]; then
  (time timeout -s KILL 900 python -m pytest tests/test_gpu_synthetic.py tests/test_input_mutations.py tests/test_flat_kernel.py -m gpu -x -q -k "$TESTS") > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
fi
for w in $WL; do
  for mode in masks staged; do
    [ $mode = staged ] && export CBH_FLAT_MASKS=0 || unset CBH_FLAT_MASKS
    BENCH="python $R/bench.py --workload $w --batches 8 --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs"
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_${w}_$mode -o r -- $BENCH > $OUT/prof_${w}_$mode.log 2>&1 )
    DB=$(find $OUT/prof_${w}_$mode -name '*.db' | head -1)
    [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" $OUT/kernel_stats_${w}_$mode.txt | sed -n 3,5p
    rm -rf $OUT/prof_${w}_$mode
    (timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs 2>$OUT/bench_${w}_$mode.err | grep '^{' | tail -1) > $OUT/bench_${w}_$mode.json
    python - <<P
import json
try:
    d = json.load(open('$OUT/bench_${w}_$mode.json')); r = d['roofline']; s = r.get('serial') or {}
    print('$w $mode', '%.3g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us frac %.3f' % (s.get('kernel_ms', 0) * 1e3, s.get('frac', 0)))
except Exception as e:
    print('$w $mode failed', e); print(open('$OUT/bench_${w}_$mode.err').read()[-1500:])
P
  done
done
unset CBH_FLAT_MASKS
