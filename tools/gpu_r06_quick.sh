#!/bin/bash
# Round 6: a quick look at the shipped build - some GPU-tier files, then the default bench line.   usage: gpu_r06_quick.sh TAG "pytest args" [bench args]
set -u
TAG=${1:-r06q}; PT=${2:-""}; BA=${3:-"--steps 20 --warmup 3"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
if [ -n "$PT" ]; then (time timeout -k 5 900 python -m pytest $PT -m gpu -q -x) > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log; fi
(timeout -k 5 600 python bench.py $BA 2>$OUT/bench.err | grep '^{' | tail -1) > $OUT/bench.json
python - <<PY
import json
d = json.load(open('$OUT/bench.json'))
r = d['roofline']; s = r.get('serial') or {}
print('value %.4g' % d['value'], 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3))
for k in sorted(d):
    if 'inclusive' in k and 'note' not in k: print(k, d[k])
print('target_T', {k: v for k, v in (d.get('target_T') or {}).items() if k in ('value', 'frac', 'serial_kernel_ms', 'kernel_ms')})
PY
tail -3 $OUT/bench.err
