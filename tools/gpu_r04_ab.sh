#!/bin/bash
# A/B of one environment switch over bench lines (no profiler).  usage: gpu_r04_ab.sh TAG "C2 C3" VAR=VALUE [extra bench args]
set -u
TAG=${1:-r04ab}; WL=${2:-"C2 C3"}; SW=${3:-CBH_FLAT_MASKS=1}; shift 3 || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for w in $WL; do
  for mode in base switch; do
    if [ $mode = switch ]; then export "$SW"; else unset "${SW%%=*}"; fi
    timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_${w}_$mode.json 2> $OUT/bench_${w}_$mode.err
    python - <<P
import json
try:
    d = json.load(open('$OUT/bench_${w}_$mode.json')); r = d['roofline']; s = r.get('serial') or {}
    print('$w $mode', '%.3g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us frac %.3f' % (s.get('kernel_ms', 0) * 1e3, s.get('frac', 0)), 'pcie %.3g' % d.get('pcie_inclusive_decisions_per_s', 0))
except Exception as e:
    print('$w $mode failed', e); print(open('$OUT/bench_${w}_$mode.err').read()[-1500:])
P
  done
done
unset "${SW%%=*}"
