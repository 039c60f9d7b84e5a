#!/usr/bin/env python3
"""Which lines of the kernels' / the library's source does NO test execute?  Union of two gcov runs: the GPU tier's bodies on the
simulator build of the library (tools/sim_engine_coverage.sh <dirA>) and the CPU tier on a --coverage build of the kernel simulation
(tests/hostsim/hostsim.cpp compiled with --coverage into <dirB>, the tier run, gcov there).  A line counts as executed when any
template instantiation in either run ran it.
    python tools/coverage_union.py <dirA> <dirB> [listing_dir]"""
import glob
import os
import re
import sys


def load(d):
    out = {}
    for f in glob.glob(os.path.join(d, "cbh_*.gcov")):
        best, src = {}, {}
        for line in open(f, errors="replace"):
            m = re.match(r"\s*([^:]+):\s*(\d+):(.*)", line)
            if not m or m.group(2) == "0" or m.group(1).strip() == "-":
                continue
            c, n = m.group(1).strip(), int(m.group(2))
            best[n] = max(best.get(n, 0), 0 if (c.startswith("#####") or c.startswith("=====")) else 1)
            src[n] = m.group(3)
        out[os.path.basename(f)[:-5]] = (best, src)
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
listing = sys.argv[3] if len(sys.argv) > 3 else None
print("%-20s %10s %18s %18s %14s" % ("source", "code lines", "not by GPU tier", "not by CPU tier", "by neither"))
for name in sorted(set(a) | set(b)):
    ba, sa = a.get(name, ({}, {}))
    bb, sb = b.get(name, ({}, {}))
    lines = set(ba) | set(bb)
    miss_a = sum(1 for n in ba if not ba[n]) if ba else "-"
    miss_b = sum(1 for n in bb if not bb[n]) if bb else "-"
    neither = sorted(n for n in lines if not ba.get(n, 0) and not bb.get(n, 0))
    print("%-20s %10d %18s %18s %14d" % (name, len(lines), miss_a, miss_b, len(neither)))
    if listing:
        os.makedirs(listing, exist_ok=True)
        open(os.path.join(listing, name + ".never.txt"), "w").write("".join("%d:%s\n" % (n, sa.get(n) or sb.get(n)) for n in neither))
