#!/bin/bash
# GPU box: the whole GPU test tier and the default bench line (the last full validation of the round)
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$PWD/gpurun_out/r03final2; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -s KILL 400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
(timeout -s KILL 200 python bench.py --steps 20 --warmup 3 2>$O/bench.err | grep '^{' | tail -1) > $O/bench_C2.json; python -c "
import json; d=json.load(open('$O/bench_C2.json')); print({k:d[k] for k in d if k.startswith('wire') or k in ('value','pcie_inclusive_decisions_per_s')}); print(d['roofline']['frac'], d['roofline']['serial']['frac'])"; tail -2 $O/bench.err
