"""Static instruction statistics of the gfx950 kernels (no GPU needed): compiles the device side to assembly and counts,
per kernel, instructions by class, SGPR-spill traffic (v_readlane / v_writelane), scratch accesses and waitcnts.
The decision kernels are instruction-issue bound, so these counts are the offline proxy for a code
change before it gets GPU time.      python tools/isa_stats.py [kernel-name-substring]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else "leaf_a4_f0"
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "cerbos_amd/csrc/cbh_engine.hip"), "-o", asm])
    text = open(asm).read()
CLASSES = [("spill_lane", r"v_(readlane|writelane)_b32"), ("scratch", r"scratch_"), ("smem", r"s_(load|buffer_load)"),
           ("vmem", r"(global|buffer|flat)_(load|store|atomic)"), ("lds", r"ds_"), ("branch", r"s_(c?branch|setpc|swappc|call)"),
           ("waitcnt", r"s_waitcnt|s_nop"), ("salu", r"s_"), ("valu", r"v_")]
cur, stats = None, {}
for line in text.splitlines():
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        stats[cur] = collections.Counter()
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
    if cur is None:
        continue
    ins = line.strip().split(" ")[0].split("\t")[0]
    if not re.match(r"^[a-z_0-9]+$", ins) or ins.startswith("."):
        continue
    for name, pat in CLASSES:
        if re.match(pat, ins):
            stats[cur][name] += 1
            break
    stats[cur]["total"] += 1
for k, c in stats.items():
    if want in k and c["total"]:
        print("%-60s total %5d | valu %5d salu %5d smem %4d vmem %4d lds %4d branch %4d waitcnt %4d | spill-lane %4d scratch %4d"
              % (k[:60], c["total"], c["valu"], c["salu"], c["smem"], c["vmem"], c["lds"], c["branch"], c["waitcnt"], c["spill_lane"], c["scratch"]))
