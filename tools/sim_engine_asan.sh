#!/bin/bash
# The GPU tier's test bodies against the simulator build of the library (tests/sim_engine.py) under AddressSanitizer: "device" memory
# is heap memory there, so a kernel (or the host code) that reads or writes outside an allocation aborts the test instead of
# passing by luck.  libstdc++ is preloaded with the runtime: its interceptor of __cxa_throw must find the real one before the
# ingest library throws (and catches) its first exception.  (ASan warns that it does not fully support swapcontext - the fibers; no false positive has shown up.)
#   tools/sim_engine_asan.sh [pytest arguments, default: tests -m gpu -q -n 8]
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/cbh_sim_asan; mkdir -p $OUT
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address -x c++ -fPIC -shared -I$R/tests/hostsim/fakehip -I$R/include \
    $R/cerbos_amd/csrc/cbh_engine.hip -o $OUT/libcerbos_hip_sim_asan.so -lpthread -ldl
cd $R
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1 \
  CBH_TEST_SIM_ENGINE=1 CBH_TEST_SIM_LIB=$OUT/libcerbos_hip_sim_asan.so python -m pytest ${@:-tests -m gpu -q -n 8 --timeout 1800 -p no:cacheprovider}
