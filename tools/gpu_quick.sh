#!/bin/bash
# quick GPU check: build, the engine tests, smoke, one short default bench
set -u
OUT=gpurun_out/${1:-quick}
mkdir -p "$OUT"
export TMPDIR=/tmp
python __graft_entry__.py > "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 1; }
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_synthetic.py -x -q -m gpu --durations=8 > "$OUT/pytest_engine.log" 2>&1; tail -30 "$OUT/pytest_engine.log"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --batches 12 > "$OUT/bench.log" 2>&1; tail -1 "$OUT/bench.log" | tee "$OUT/bench.json" | cut -c1-1800
