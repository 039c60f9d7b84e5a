#!/bin/bash
# One gpurun call's worth of measurements, cheapest first (run on the GPU box from the repository root):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_all.sh r02'
# Writes everything under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-rNN}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 200 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; tail -2 "$OUT/pytest_gpu.log"
CBH_GPU_FUZZ_WIDE=1 timeout 60 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > "$OUT/pytest_gpu_fuzz_wide.log" 2>&1; tail -1 "$OUT/pytest_gpu_fuzz_wide.log"
timeout 90 python bench.py > "$OUT/bench.log" 2>&1; tail -1 "$OUT/bench.log" > "$OUT/bench.json"; cut -c1-300 "$OUT/bench.json"
for w in C1 C3 C4 C5; do timeout 120 python bench.py --workload $w --steps 100 --warmup 20 2>/dev/null | tail -1 > "$OUT/bench_$w.json"; done
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/rocprof" -- python "$OLDPWD/bench.py" --steps 200 --warmup 20 --no-cpu-baseline > "$OLDPWD/$OUT/rocprof.log" 2>&1 )
timeout 60 python tools/e2e_bench.py C2 131072 8192 2 1,8,32,64 > "$OUT/e2e_outer.log" 2>&1; grep '^{"threads"' "$OUT/e2e_outer.log"
timeout 60 python tools/e2e_bench.py C2 131072 32768 2 2,4,8 16 > "$OUT/e2e_inner16.log" 2>&1; grep '^{"threads"' "$OUT/e2e_inner16.log"
timeout 90 python tools/ingest_bench.py C2 200000 > "$OUT/ingest_bench.log" 2>&1; tail -12 "$OUT/ingest_bench.log"
# the same end-to-end loop without Python in it (no GIL between the C calls)
python tools/export_wire.py C2 131072 /tmp/c2wire > /dev/null && g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$PWD/cerbos_amd -o /tmp/e2e_bench \
  && timeout 90 /tmp/e2e_bench /tmp/c2wire 8192 2 1,8,32,64,128,224 > "$OUT/e2e_cpp.log" 2>&1; cat "$OUT/e2e_cpp.log"
timeout 120 python tools/cross_bench.py C2 2000 1000 50 > "$OUT/cross_bench.log" 2>&1; tail -1 "$OUT/cross_bench.log"
