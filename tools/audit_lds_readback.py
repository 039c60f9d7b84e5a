#!/usr/bin/env python3
"""Offline audit (no GPU) of the one hardware-only bug round 5 met: wave_or64 (cbh_check_flat.h) ORs the lanes' words into an LDS
word under `if (v)` and reads the word back; with a plain read the compiler had moved the read INTO that branch, so lanes that OR
nothing in never read (profiles/r05_walk2_candidate_filter_ab.txt).  This compiles the device side to assembly and checks, for
every kernel, that each ds_or_b64 is followed by its ds_read_b64 only AFTER the exec mask was restored (s_or_b64 exec, ...).
    python tools/audit_lds_readback.py [asm-file]        exit status 1 if a read sits inside the branch"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = sys.argv[1] if len(sys.argv) > 1 else None
if asm is None:
    asm = os.path.join(tempfile.gettempdir(), "cbh_audit.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "cerbos_amd/csrc/cbh_engine.hip"), "-o", asm])
cur, kern = None, {}
for line in open(asm).read().splitlines():
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1); kern[cur] = []
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
    if cur:
        kern[cur].append(line)
bad = checked = 0
for k, lines in kern.items():
    for i, l in enumerate(lines):
        if "ds_or_b64" not in l:
            continue
        restored = False
        for j in range(i + 1, min(i + 400, len(lines))):
            if "s_or_b64 exec" in lines[j]:
                restored = True
            if "ds_read_b64" in lines[j]:
                checked += 1
                if not restored:
                    bad += 1
                    print("READ INSIDE THE BRANCH:", k[:70], "line", j)
                break
print("%d read-backs checked, %d inside their branch" % (checked, bad))
sys.exit(1 if bad else 0)
