#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/probe
python __graft_entry__.py > gpurun_out/probe/build.log 2>&1
run() { echo "== $*"; env "$@" CBH_TRACE=1 python tools/gpu_probe_oneshot.py 2> gpurun_out/probe/trace.err; grep '\[cbh\] range' gpurun_out/probe/trace.err | tail -2; grep '\[cbh\] small' gpurun_out/probe/trace.err | awk 'NR%300==150'; }
run A=default
run CBH_COPY_MODE=1
run CBH_CHUNK_REQUESTS=250048
run CBH_CHUNK_REQUESTS=32768
run CBH_ZEROCOPY_BYTES=0
run CBH_SPIN=1
run CBH_SPIN=1 CBH_ZEROCOPY_BYTES=0
timeout 300 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3
