set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$PWD/gpurun_out/r03final3; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -s KILL 120 python -m pytest tests/test_gpu_wire.py -m gpu -x -q > $O/pytest_wire.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_wire.log
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$PWD/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
python tools/export_wire.py C2 262144 /tmp/wire_C2 > $O/export_C2.log 2>&1
timeout -s KILL 40 /tmp/e2e_wire_bench /tmp/wire_C2 131072 1.5 1,4 device_out | tee $O/e2e_C2.json | grep road
CBH_WIRE_GROUP=0 timeout -s KILL 40 /tmp/e2e_wire_bench /tmp/wire_C2 131072 1.5 1 device_out | tee $O/e2e_C2_nogroup.json | grep road
