set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$PWD/gpurun_out/r03final; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
(time timeout -s KILL 500 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest_gpu.log
(time bash tools/gpu_final_r03.sh r03final) 2>&1 | tail -12
(time bash tools/gpu_r03_wire.sh r03final notests "C2 C5") 2>&1 | grep -E "road|verified|real|wire_" | cut -c1-200
