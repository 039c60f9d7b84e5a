#!/bin/bash
# Which lines of the kernels' and the library's source do the GPU tier's test bodies execute?  The simulator build of the library
# (tests/sim_engine.py) compiled with --coverage, the tier run against it, gcov per source file -> a table and the .gcov listings.
#   tools/sim_engine_coverage.sh [out_dir]      (about twenty minutes with eight workers: the build is -O0)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-${TMPDIR:-/tmp}/cbh_sim_cov}; mkdir -p $OUT; rm -f $OUT/*.gc* $OUT/*.gcov
g++ -std=c++17 -O0 -g --coverage -x c++ -fPIC -c -I$R/tests/hostsim/fakehip -I$R/include $R/cerbos_amd/csrc/cbh_engine.hip -o $OUT/engine.o
g++ -shared --coverage $OUT/engine.o -o $OUT/libcerbos_hip_sim_cov.so -lpthread -ldl
cd $R
CBH_TEST_SIM_ENGINE=1 CBH_TEST_SIM_LIB=$OUT/libcerbos_hip_sim_cov.so python -m pytest tests -m gpu -q -n 8 --timeout 1500 -p no:cacheprovider | tail -2
cd $OUT && gcov -o $OUT engine.gcno > gcov.log 2>&1
python3 - <<'PY'
import glob, re
print("%-22s %10s %14s %8s" % ("source", "code lines", "never executed", "executed"))   # (a line counts when ANY template instantiation ran it)
for f in sorted(glob.glob("cbh_*.gcov")):
    best = {}
    for line in open(f, errors="replace"):
        m = re.match(r"\s*([^:]+):\s*(\d+):(.*)", line)
        if not m or m.group(2) == "0" or m.group(1).strip() == "-":
            continue
        c, n = m.group(1).strip(), int(m.group(2))
        best[n] = max(best.get(n, 0), 0 if (c.startswith("#####") or c.startswith("=====")) else 1)
    miss = sum(1 for v in best.values() if not v)
    print("%-22s %10d %14d %7.1f%%" % (f[:-5], len(best), miss, 100.0 * (len(best) - miss) / max(1, len(best))))
PY
