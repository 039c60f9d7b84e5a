#!/bin/bash
# GPU box: test tier + a short bench line per workload (no profiler). usage: gpu_quick_r03.sh TAG "C2 C5 T" [notests]
set -u
TAG=${1:-r03q}; WL=${2:-"C2 C5"}; NOTESTS=${3:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ -z "$NOTESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_gpu.log
fi
for w in $WL; do
  (timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>$OUT/bench_$w.err | grep '^{' | tail -1) > $OUT/bench_$w.json
  python - <<P
import json
try:
    d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']; s=r.get('serial') or {}
    print('$w', '%.3g dec/s' % d['value'], r['kernel'], 'kernel %.1f us' % (r['kernel_ms']*1e3), 'frac %.3f' % r['frac'], 'serial %.1f us frac %.3f' % (s.get('kernel_ms',0)*1e3, s.get('frac',0)), 'pcie %.3g' % d.get('pcie_inclusive_decisions_per_s',0))
except Exception as e:
    print('$w failed', e); print(open('$OUT/bench_$w.err').read()[-1500:])
P
done
