#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average
duration plus launch geometry and register use.  Usage: rocpd_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    lines = ["# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1].split("/")[-1],
             "%-70s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, total, avg, pct in cur.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"):
        lines.append("%-70s %8d %14.3f %12.3f %6.2f%%" % (name[:70], calls, total, avg, pct))
    lines.append("")
    lines.append("%-40s %10s %6s %8s %6s %6s %9s" % ("kernel", "grid_x", "wg_x", "lds", "vgpr", "sgpr", "scratch"))
    seen = set()
    for row in cur.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size from kernels"):
        if row[0] in seen:
            continue
        seen.add(row[0])
        lines.append("%-40s %10d %6d %8d %6d %6d %9d" % ((row[0][:40],) + tuple(row[1:])))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
