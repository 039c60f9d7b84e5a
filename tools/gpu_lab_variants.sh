#!/bin/bash
# build the library with each CBH_EXTRA_FLAGS variant and time the resident C2 bench (kernel lab)
export TMPDIR=/tmp
mkdir -p gpurun_out/variants
for v in "$@"; do
  CBH_EXTRA_FLAGS="$v" bash tools/build_all.sh > gpurun_out/variants/build.log 2>&1 || { echo "build failed: $v"; tail -5 gpurun_out/variants/build.log; continue; }
  touch cerbos_amd/libcerbos_hip.so
  r=$(timeout 200 python bench.py --workload C2 --batches 12 --steps 10 --warmup 2 --no-cpu-baseline --no-side-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('kernel_ms %.4f frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))")
  echo "variant [$v]: $r"
done
