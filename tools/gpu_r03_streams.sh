#!/bin/bash
# GPU box: resident-stream sweep of one workload (bench.py without side legs)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-r03_streams}; W=${2:-C2}; SW=${3:-"1 2 3 4 6 8"}
mkdir -p $OUT; cd $R
python __graft_entry__.py > $OUT/build.log 2>&1
for s in $SW; do
  CBH_RESIDENT_STREAMS=$s timeout 300 python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_${W}_s$s.json
  python -c "
import json; d=json.load(open('$OUT/bench_${W}_s$s.json')); r=d['roofline']; print('$W streams', r['streams'], '%.4g dec/s' % d['value'], 'kernel %.1f us' % (r['kernel_ms']*1e3), 'frac %.3f' % r['frac'])"
done
