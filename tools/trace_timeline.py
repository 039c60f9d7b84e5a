#!/usr/bin/env python3
"""A rocprofv3 trace directory (--kernel-trace --memory-copy-trace --hip-trace, csv) -> the timeline of the LAST call of the traced
program: every copy and kernel with start / end relative to the window's first event, per stream / agent, plus how much of the window
each direction of the link and the kernels were busy.   python tools/trace_timeline.py DIR [window_ms]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0], r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "")), r.get("Stream_Id", "")))
ev.sort()
if not ev:
    sys.exit("no events")
end = ev[-1][1]
start = end - int(win_ms * 1e6)
w = [e for e in ev if e[0] >= start]
# the window's first big H2D marks the call's beginning
t0 = w[0][0]
busy = {}
for s, e, k, name, st in w:
    key = name if k == "C" else "kernels"
    busy.setdefault(key, []).append((s, e))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
print("window %.3f ms, %d events" % ((end - t0) / 1e6, len(w)))
for k, iv in busy.items():
    print("  busy %-28s %.3f ms in %d events" % (k, union(iv) / 1e6, len(iv)))
for s, e, k, name, st in w:
    if e - s >= 20000 or k == "C":
        print("%9.1f us  +%8.1f us  %s %-40s stream %s" % ((s - t0) / 1e3, (e - s) / 1e3, k, name[:40], st))
