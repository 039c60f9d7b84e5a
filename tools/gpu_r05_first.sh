#!/bin/bash
# The first GPU call of the next round: everything round 4 wrote after its GPU minutes were spent, measured - and the artefacts
# of the last build whole (round 4's bench lines of C3 .. C5W and its PMC traffic are an earlier build's).
#   1. the GPU tier with the files that sort last (request road, effective policies)
#   2. tools/gpu_final_r04.sh (kernel statistics, PMC traffic - C3 with the tags as bytes -, bench lines, all on this build)
#   3. tools/gpu_requests_and_trail.py: the request road against the input road, the trail walk against cbh_check_batch
set -u
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
tools/gpu_final_r04.sh $TAG
# 4. the pre-pass in two kernels against the fused one
for mode in 0 1; do
  for w in C5 C5W; do
    CBH_PRE_SPLIT=$mode CBH_BENCH_NO_DIST=1 timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_${w}_presplit$mode.json 2> $OUT/bench_${w}_presplit$mode.err
    python -c "
import json; d=json.load(open('$OUT/bench_${w}_presplit$mode.json')); r=d['roofline']; s=r.get('serial') or {}
print('$w CBH_PRE_SPLIT=$mode', '%.3g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3))"
  done
done
CBH_PRE_SPLIT=1 timeout 300 python -m pytest tests/test_gpu_synthetic.py -m gpu -x -q -k "c5 or C5" 2>&1 | tail -3
for w in C2 C5; do
  timeout 300 python tools/gpu_requests_and_trail.py $w 250000 10 > $OUT/requests_and_trail_$w.txt 2>&1; tail -8 $OUT/requests_and_trail_$w.txt
done
# 5. what decision logs cost: the trail forms of the kernels, resident (bench.py --audit-trail), against the lines of step 2
for w in C2 C3 T C5 C5W; do
  CBH_BENCH_NO_DIST=1 timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --audit-trail > $OUT/bench_${w}_trail.json 2> $OUT/bench_${w}_trail.err
  python -c "
import json; d=json.load(open('$OUT/bench_${w}_trail.json')); r=d['roofline']; s=r.get('serial') or {}
print('$w with the audit trail', '%.3g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3))"
done
