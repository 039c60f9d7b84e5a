#!/bin/bash
# kernel lab: resident-rate benches of the four configurations, kernel time only (no side legs, no CPU baseline)
#   bash tools/gpu_lab.sh <tag> [workloads...]
set -u
TAG=${1:-lab}; shift || true
WLS=${*:-C2 C3 C4 C5}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python __graft_entry__.py > "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 1; }
for w in $WLS; do
  NB=$(case $w in C2) echo 12;; C3) echo 2;; C4) echo 3;; C5) echo 6;; *) echo 16;; esac)
  timeout 300 python bench.py --workload $w --batches $NB --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs > "$OUT/bench_$w.log" 2>&1
  tail -1 "$OUT/bench_$w.log" > "$OUT/bench_$w.json"
  python - "$OUT/bench_$w.json" $w <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%s %-32s kernel_ms %.4f  frac %.4f  value %.3g/s  allow %.3f" % (sys.argv[2], r["kernel"], r["kernel_ms"], r["frac"], d["value"], d["allow_fraction"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
    print(open(sys.argv[1].replace(".json", ".log")).read()[-1500:])
PY
done
