set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03w3; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
(timeout 600 python bench.py --steps 20 --warmup 3 2>$O/bench.err | grep '^{' | tail -1) > $O/bench_C2.json; python -c "
import json; d=json.load(open('$O/bench_C2.json')); print({k:d[k] for k in d if k.startswith('wire') or k in ('value','pcie_inclusive_decisions_per_s')}); print(d['roofline']['frac'], d['roofline']['serial']['frac'])"; tail -3 $O/bench.err
g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$PWD/cerbos_amd -o /tmp/e2e_wire_bench || exit 1
python tools/export_wire.py C2 524288 /tmp/wire_C2 > $O/export.log 2>&1
for S in 16384 131072; do timeout 200 /tmp/e2e_wire_bench /tmp/wire_C2 $S 2 1,4,8,16,32 device_out > $O/e2e_C2_$S.json 2>$O/e2e_C2_$S.err; cat $O/e2e_C2_$S.json; tail -2 $O/e2e_C2_$S.err; done
