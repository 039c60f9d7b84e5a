#!/usr/bin/env python3
"""Turns the rocprofv3 counter CSVs of tools/gpu_final_r03.sh into pmc_calibration.json and pmc_traffic.json.

Calibration (tools/pmc_calib.hip moves exactly 512 MiB per launch with one access shape): factor = known bytes /
reported bytes per shape.  Traffic of the decision kernels = FETCH_SIZE x factor(dword reads) + WRITE_SIZE x
factor(dword writes): the kernels read with 4 B/lane loads and LDS-DMA dwords and write dwords / 16-byte vectors.
Usage: pmc_summary.py gpurun_out/<tag>"""
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]


def counters(pattern, want):
    vals = {}
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name", "").split("(")[0]
                if want(name):
                    vals.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return vals


KNOWN = 512 << 20
cal = counters(os.path.join(out_dir, "pmc_calib", "**", "*counter_collection.csv"), lambda n: n.startswith("calib_"))
calib = {}
for (name, ctr), v in sorted(cal.items()):
    relevant = (ctr == "FETCH_SIZE") == ("read" in name)
    if not relevant:
        continue
    kb = sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0]   # first launch of a buffer may hit memset leftovers
    calib[name] = {"counter": ctr, "reported_kb": kb, "known_bytes": KNOWN, "factor": KNOWN / (kb * 1024.0) if kb else None, "launches": len(v)}
json.dump(calib, open(os.path.join(out_dir, "pmc_calibration.json"), "w"), indent=1)
print(json.dumps(calib))

f_read = (calib.get("calib_read_b32") or {}).get("factor") or 1.0
f_lds = (calib.get("calib_read_lds_b32") or {}).get("factor") or f_read
f_write = (calib.get("calib_write_b32") or {}).get("factor") or 1.0
traffic = {"calibration": {"read_b32": f_read, "read_lds_b32": f_lds, "write_b32": f_write,
                           "source": "tools/pmc_calib.hip, 512 MiB per launch, same box and profiler"}, "workloads": {}}
for d in sorted(glob.glob(os.path.join(out_dir, "pmc_*"))):
    w = os.path.basename(d)[4:]
    if w == "calib" or not os.path.isdir(d):
        continue
    # a batch is decided by one launch of each kernel of its plan (cbh_walk2_pre_kernel + cbh_walk2_kernel, or one flat /
    # general kernel): the per-batch traffic is the sum of their per-launch averages
    vals = counters(os.path.join(d, "**", "*counter_collection.csv"), lambda n: n.startswith("cbh_check_") or n.startswith("cbh_walk2"))
    kernels = sorted({k for k, _ in vals})
    if not kernels:
        continue
    fetch_kb = write_kb = 0.0
    launches = 0
    for k in kernels:
        fk = vals.get((k, "FETCH_SIZE")) or [0.0]
        wk = vals.get((k, "WRITE_SIZE")) or [0.0]
        fetch_kb += sum(fk) / len(fk); write_kb += sum(wk) / len(wk)
        launches = max(launches, len(fk))
    traffic["workloads"][w] = {
        "kernel": "+".join(kernels), "kernels": kernels, "fetch_kb_per_launch_raw": fetch_kb, "write_kb_per_launch_raw": write_kb,
        "bytes_per_launch": fetch_kb * 1024.0 * f_read + write_kb * 1024.0 * f_write, "launches": launches,
        "note": "FETCH_SIZE x %.3f + WRITE_SIZE x %.3f (factors calibrated on 4 B/lane accesses of known size); rotating set, "
                "bench.py --steps 6 --warmup 2" % (f_read, f_write)}
json.dump(traffic, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic["workloads"]))
