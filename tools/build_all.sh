#!/bin/bash
# Rebuild the host simulator + the gfx950 library; print the kernels' resource usage.
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -std=c++17 -shared -fPIC -Iinclude tests/hostsim/hostsim.cpp -o tests/hostsim/libcbh_hostsim.so
g++ -O2 -std=c++17 -Wall -Wextra -shared -fPIC -pthread -Iinclude cerbos_amd/csrc/cbh_ingest.cpp -o cerbos_amd/libcerbos_ingest.so
cd cerbos_amd/csrc
LOG=${CBH_BUILD_LOG:-$(mktemp)}
# CBH_PROFILE=1 builds the per-wave cycle counters in (tools/gpu_cycles.py); never ship that build.
if ! hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC ${CBH_PROFILE:+-DCBH_PROFILE_CYCLES} ${CBH_ABLATION:+-DCBH_ABLATION} ${CBH_EXTRA_FLAGS:-} -I../../include cbh_engine.hip -o ../libcerbos_hip.so -Rpass-analysis=kernel-resource-usage > "$LOG" 2>&1; then
  grep -E "error" -A3 "$LOG" | head -40
  echo "HIPCC FAILED"
  exit 1
fi
grep -E "Function Name|VGPRs:|Occupancy|SGPRs Spill|Scratch|LDS Size" "$LOG" | grep -A5 "check_kernel" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste -sd' ' | sed 's/Function Name/\nFunction Name/g'
echo
