#!/bin/bash
# Round 6: N builds of the library on ONE box (ab_<name>.so at the repo root), the workloads' resident lines in turn, twice; then
# (optionally) GPU-tier tests on the LAST named build.   usage: gpu_r06_ab.sh TAG "old new ..." "C2 T" [pytest args]
set -u
TAG=${1:-r06ab}; VARS=${2:-"old new"}; WL=${3:-"C2 T"}; PT=${4:-""}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp CBH_BENCH_NO_DIST=1
B="--steps 10 --warmup 2 --no-cpu-baseline --no-side-legs --serial-leg --check-first"
LAST=""
for rep in 1 2; do
  for v in $VARS; do
    cp ab_$v.so cerbos_amd/libcerbos_hip.so; LAST=$v
    for w in $WL; do
      timeout -k 5 200 python bench.py --workload $w $B > $OUT/bench_${w}_${v}_$rep.json 2> $OUT/bench_${w}_${v}_$rep.err
      python -c "
import json
try:
    d=json.loads([l for l in open('$OUT/bench_${w}_${v}_$rep.json') if l.startswith('{')][-1]); r=d['roofline']; s=r.get('serial') or {}
    print('%-4s %-10s #$rep' % ('$w', '$v'), '%.4g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'kernel %.1f us' % (r.get('kernel_ms', 0) * 1e3), 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3), '| ' + str(d.get('first_batch_checked')))
except Exception as e: print('$w $v FAILED', e)"
    done
  done
done 2>&1 | tee $OUT/ab.txt
if [ -n "$PT" ]; then timeout -k 5 900 python -m pytest $PT -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest ($LAST) rc $?"; tail -3 $OUT/pytest.log; fi
