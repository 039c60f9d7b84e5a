#!/bin/bash
# Round 5: a wider differential fuzz on the hardware (seeds beyond the test tier's), bounded by its own clock.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05z
timeout -k 5 260 python tools/gpu_fuzz_sweep.py 700000 100000 150 2>&1 | tail -3 | tee gpurun_out/r05z/fuzz_sweep.txt
timeout -k 5 120 python tools/gpu_fuzz_sweep.py 800000 100000 60 2>&1 | tail -1 | tee -a gpurun_out/r05z/fuzz_sweep.txt
