#!/bin/bash
# round 2, call A: the new GPU-tier parity tests + baseline rocprofv3 kernel stats of C3/C4/C5 (round-1 kernels)
set -u
OUT=gpurun_out/r02a
mkdir -p "$OUT"
export TMPDIR=/tmp
python __graft_entry__.py > "$OUT/build.log" 2>&1
timeout 900 python -m pytest tests -x -q -m gpu --durations=15 > "$OUT/pytest_gpu.log" 2>&1; tail -25 "$OUT/pytest_gpu.log"
for w in C3 C4 C5; do
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/rocprof_$w" -- python "$OLDPWD/bench.py" --workload $w --steps 50 --warmup 10 --no-cpu-baseline > "$OLDPWD/$OUT/rocprof_$w.log" 2>&1 )
  tail -1 "$OUT/rocprof_$w.log" | cut -c1-400
  f=$(find "$OUT/rocprof_$w" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -5 "$f"
done
