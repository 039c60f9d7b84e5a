#!/usr/bin/env python3
"""Where a kernel's instructions come from (no GPU needed): the device side compiled to assembly with line tables
(hipcc ... --cuda-device-only -S -gline-tables-only), every instruction of one kernel attributed to the source FUNCTION its line
lies in.  For the straight-line, fully inlined kernels here (the wire kernels execute most of their code about once per wave) the
static count is the offline proxy for where the issue slots go.
    python tools/isa_by_source.py <kernel-name-substring> [asm-file]     (asm-file: reuse an earlier compile)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1]
asm = sys.argv[2] if len(sys.argv) > 2 else None
if asm is None:
    asm = os.path.join(tempfile.gettempdir(), "cbh_lines.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-gline-tables-only",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "cerbos_amd/csrc/cbh_engine.hip"), "-o", asm])
files = {}
text = open(asm).read().splitlines()
for l in text:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = m.group(2)
# function spans per source file: "name(" at the start of a definition, by a crude scan (good enough for these headers)
spans = {}
for fid, name in files.items():
    path = os.path.join(ROOT, "cerbos_amd/csrc", name)
    if not os.path.exists(path):
        continue
    cur, out = "?", []
    for n, line in enumerate(open(path, errors="replace").read().splitlines(), 1):
        m = re.match(r'^(?:template\s*<[^>]*>\s*)?(?:__device__|__global__|static|inline|__forceinline__|__attribute__\(\([^)]*\)\)|\s)+[\w:<>\*&\s]+?\b(\w+)\s*\(', line)
        if m and not line.startswith(" ") and m.group(1) not in ("if", "for", "while", "switch", "return", "sizeof", "defined"):
            cur = m.group(1)
        out.append(cur)
    spans[fid] = out
cur_kernel, fid, line = None, 0, 0
by_fn = collections.Counter(); by_cls = collections.defaultdict(collections.Counter); total = 0
for l in text:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur_kernel = m.group(1) if want in m.group(1) else None
        continue
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
        cur_kernel = None
    if cur_kernel is None:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        fid, line = int(m.group(1)), int(m.group(2))
        continue
    ins = l.strip().split(" ")[0].split("\t")[0]
    if not re.match(r'^[a-z_0-9]+$', ins) or ins.startswith("."):
        continue
    sp = spans.get(fid)
    fn = "%s:%s" % (files.get(fid, "?"), sp[line - 1] if sp and 0 < line <= len(sp) else "?")
    cls = "spill" if ins.startswith(("v_readlane", "v_writelane")) else "valu" if ins.startswith("v_") else "salu" if ins.startswith("s_") else "mem"
    by_fn[fn] += 1; by_cls[fn][cls] += 1; total += 1
print("%s: %d instructions" % (want, total))
for fn, n in by_fn.most_common(28):
    c = by_cls[fn]
    print("  %6d  %4.1f %%  %-52s valu %5d  salu %5d  spill %5d  mem %4d" % (n, 100.0 * n / total, fn, c["valu"], c["salu"], c["spill"], c["mem"]))
