#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage remarks (stderr of a build) -> one line per kernel.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> res.txt; python tools/kernel_resources.py res.txt [filter]"""
import re
import sys

cur, rows = None, []
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"remark: (?:\S+: )?Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        d = re.match(r"_Z(\d+)", name)
        if d:
            n = int(d.group(1)); name = name[2 + len(d.group(1)):][:n]
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?(\w[\w /\[\]]*?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
print("%-42s %5s %5s %6s %6s %8s %4s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "sspill", "scratch", "occ", "LDS"))
for r in rows:
    if flt in r["name"]:
        print("%-42s %5d %5d %6d %6d %8d %4d %6d" % (r["name"], r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("TotalSGPRs", 0), r.get("SGPRs Spill", 0),
                                                  r.get("ScratchSize [bytes/lane]", 0), r.get("Occupancy [waves/SIMD]", 0), r.get("LDS Size [bytes/block]", 0)))
