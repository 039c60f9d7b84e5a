#!/bin/bash
# GPU box, round 3: the GPU test tier, then the decision kernels' time per workload (cbh_walk2_kernel on C5; CBH_NO_WALK2=1 = the general walk beside it)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-r03a}
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests -x -q -m gpu --durations=10 > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
for w in C5 C3 C4 C2; do
  (timeout 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs 2>$OUT/bench_$w.err | grep '^{' | tail -1) > $OUT/bench_$w.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$w.json"))
    print("$w", d["value"], d.get("roofline"), d.get("kernel"))
except Exception as e:
    print("$w failed", e); print(open("$OUT/bench_$w.err").read()[-1500:])
PY
done
(CBH_NO_WALK2=1 timeout 300 python bench.py --workload C5 --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs 2>/dev/null | grep '^{' | tail -1) > $OUT/bench_C5_general.json
python -c "
import json; d=json.load(open('$OUT/bench_C5_general.json')); print('C5 general walk', d['value'], d.get('roofline'))"
