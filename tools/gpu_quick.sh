#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, then a short bench line.
mkdir -p gpurun_out
(timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/pytest_gpu.log
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1) > gpurun_out/bench_quick.json
cat gpurun_out/pytest_gpu.log
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_quick.json'))
print("value %.3e dec/s  ms/step %.4f  kernel_ms %.4f  frac %.4f  oneshot %.3e" % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['oneshot_pcie_inclusive_decisions_per_s']))
PY
