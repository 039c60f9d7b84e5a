#!/bin/bash
# Round 5: the trace pass's new value form on the hardware - the GPU tier's trace / golden / engine tests on the final build.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05t
python __graft_entry__.py > gpurun_out/r05t/build.log 2>&1 || { tail -5 gpurun_out/r05t/build.log; exit 1; }
timeout -k 5 400 python -m pytest tests -m gpu -q -k "trace or errors_and_outputs or golden or engine or kats or service" > gpurun_out/r05t/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r05t/pytest.log
timeout -k 5 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
