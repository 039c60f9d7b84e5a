#!/bin/bash
# Round 5: two builds of the library on ONE box (ab_old.so / ab_new.so at the repo root), the workloads' resident lines in turn, twice;
# then the GPU tier's tests of what changed on the new build.   usage: gpu_r05_libab.sh TAG "C5 C5W" [pytest args]
set -u
TAG=${1:-r05ab}; WL=${2:-"C5 C5W"}; PT=${3:-""}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp CBH_BENCH_NO_DIST=1
B="--steps 10 --warmup 2 --no-cpu-baseline --no-side-legs"
for rep in 1 2; do
  for v in old new; do
    cp ab_$v.so cerbos_amd/libcerbos_hip.so
    for w in $WL; do
      timeout -k 5 200 python bench.py --workload $w $B > $OUT/bench_${w}_${v}_$rep.json 2> $OUT/bench_${w}_${v}_$rep.err
      python -c "
import json
try:
    d=json.loads([l for l in open('$OUT/bench_${w}_${v}_$rep.json') if l.startswith('{')][-1]); r=d['roofline']; s=r.get('serial') or {}
    print('$w $v #$rep', '%.4g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3))
except Exception as e: print('$w $v FAILED', e)"
    done
  done
done 2>&1 | tee $OUT/ab.txt
cp ab_new.so cerbos_amd/libcerbos_hip.so
if [ -n "$PT" ]; then timeout -k 5 600 python -m pytest $PT -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log; fi
