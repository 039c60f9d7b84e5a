#!/bin/bash
# Round 5, last call: the GPU tier on the final build, the default bench line and the side lines that could have moved (wave_or64's
# accesses are atomic now: every decision kernel was rebuilt).   usage: gpu_r05_verify.sh TAG
set -u
TAG=${1:-r05v}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
(time timeout -k 5 600 python -m pytest tests -m gpu -q) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(timeout -k 5 600 python bench.py --steps 20 --warmup 3 2>$OUT/bench_C2.err | grep '^{' | tail -1) > $OUT/bench_C2.json
cut -c1-200 $OUT/bench_C2.json; echo
for w in C3 C4 C5 T; do
  (CBH_BENCH_NO_DIST=1 timeout -k 5 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs 2>$OUT/bench_$w.err | grep '^{' | tail -1) > $OUT/bench_$w.json
  python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']; s=r.get('serial') or {}
print('$w', '%.4g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'])"
done
