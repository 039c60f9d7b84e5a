#!/bin/bash
# Round 5, side measurements on this build (every step bounded):
#   1. resident launches on 8 streams instead of 4 (CBH_RESIDENT_STREAMS)                      -> bench_W_streams8.json
#   2. T with the column cache's tags as bytes (CBH_PACKED_TAGS=1; the launch picks dwords)     -> bench_T_packed.json
#   3. the pre-pass fused (CBH_PRE_SPLIT=0) against the default collect + interpret             -> bench_{C5,C5W}_presplit0.json
#   4. what AuditTrail.EffectivePolicies costs the resident kernels (bench.py --audit-trail)    -> bench_{C2,T,C5}_trail.json
#   5. the request road against the input road, with and without every request's trail         -> requests_and_trail_{C2,C5}.txt
#   6. a wider differential fuzz on the hardware (seeds beyond the test tier's)                 -> fuzz_sweep.txt
set -u
TAG=${1:-r05x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp CBH_BENCH_NO_DIST=1
line() {   # label file
  python -c "
import json,sys
try:
    d=json.loads([l for l in open('$2') if l.startswith('{')][-1]); r=d['roofline']; s=r.get('serial') or {}
    print('$1', '%.4g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us' % (s.get('kernel_ms', 0) * 1e3))
except Exception as e: print('$1 FAILED', e)"
}
B="--steps 10 --warmup 2 --no-cpu-baseline --no-side-legs"
for w in C2 C3 C4 C5 T; do
  CBH_RESIDENT_STREAMS=8 timeout -k 5 200 python bench.py --workload $w $B > $OUT/bench_${w}_streams8.json 2> $OUT/bench_${w}_streams8.err; line "$w streams=8" $OUT/bench_${w}_streams8.json
done
CBH_PACKED_TAGS=1 timeout -k 5 200 python bench.py --workload T $B > $OUT/bench_T_packed.json 2> $OUT/bench_T_packed.err; line "T packed tags" $OUT/bench_T_packed.json
for w in C5 C5W; do
  CBH_PRE_SPLIT=0 timeout -k 5 200 python bench.py --workload $w $B > $OUT/bench_${w}_presplit0.json 2> $OUT/bench_${w}_presplit0.err; line "$w CBH_PRE_SPLIT=0" $OUT/bench_${w}_presplit0.json
done
for w in C2 T C5; do
  timeout -k 5 200 python bench.py --workload $w $B --audit-trail > $OUT/bench_${w}_trail.json 2> $OUT/bench_${w}_trail.err; line "$w with the audit trail" $OUT/bench_${w}_trail.json
done
for w in C2 C5; do
  timeout -k 5 200 python tools/gpu_requests_and_trail.py $w 250000 10 > $OUT/requests_and_trail_$w.txt 2>&1; tail -8 $OUT/requests_and_trail_$w.txt | cut -c1-200
done
timeout -k 5 150 python tools/gpu_fuzz_sweep.py 700000 400 > $OUT/fuzz_sweep.txt 2>&1; tail -3 $OUT/fuzz_sweep.txt | cut -c1-300
