#!/bin/bash
# A/B of two builds of the library over bench lines.  usage: gpu_r04_libab.sh TAG "T C4" build_variants/lib_x.so [bench args]
set -u
TAG=${1:-r04libab}; WL=${2:-"T"}; ALT=$3; shift 3 || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
cp cerbos_amd/libcerbos_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for w in $WL; do
  for mode in base alt; do
    [ $mode = alt ] && cp $ALT cerbos_amd/libcerbos_hip.so || cp /tmp/lib_orig.so cerbos_amd/libcerbos_hip.so
    timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_${w}_$mode.json 2> $OUT/bench_${w}_$mode.err
    python - <<P
import json
try:
    d = json.load(open('$OUT/bench_${w}_$mode.json')); r = d['roofline']; s = r.get('serial') or {}
    print('$w $mode', '%.3g dec/s' % d['value'], r['kernel'], 'frac %.3f' % r['frac'], 'by itself %.1f us frac %.3f' % (s.get('kernel_ms', 0) * 1e3, s.get('frac', 0)))
except Exception as e:
    print('$w $mode failed', e); print(open('$OUT/bench_${w}_$mode.err').read()[-800:])
P
  done
done
done
cp /tmp/lib_orig.so cerbos_amd/libcerbos_hip.so
