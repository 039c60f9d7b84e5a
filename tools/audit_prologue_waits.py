#!/usr/bin/env python3
"""Where does a wave of a decision kernel wait for EVERYTHING it has in flight before it reads its first record?  (No GPU needed.)

Round 6 found the flat kernels' prologue - "two memory round trips" by its source - waiting with `s_waitcnt vmcnt(0)` at three to twelve
places: a load inside a conditional block is waited for at the block's end, an LDS access behind an asynchronous
global->LDS copy waits for all copies, a load and its LDS store in one loop body are a round trip per iteration.  This script reads the
SHIPPED library the way that was found - the code object out of the .so, disassembled - and prints, per kernel, the vector-memory
instructions and the full waits between the kernel's entry and its first `s_barrier` (the prologue's end in the flat kernels and the
walk), as a sequence of round trips: a trip = the loads issued since the last full wait.  The scan is STATIC - it follows the code's order, not a
wave's path: blocks that exclude each other (packed / wide tags, the fallbacks behind a missed speculation, the tail loops for large
tables, the per-lane loop that climbs to a scope with a policy) all show up; what to look for is a trip that carries bulk loads
(lds-copy, the request words) behind an earlier one that did too.

    python tools/audit_prologue_waits.py [kernel-name-substring ...]        (default: the flat kernel, the mask walk, the walk)
    python tools/audit_prologue_waits.py --lib path/to/libcerbos_hip.so ...

Exit status 1 if a listed kernel's prologue has more than --max-bulk-trips (default 3) BULK trips - trips that carry asynchronous copies
or at least eight loads of a dword or more: the request words with everything else that depends on nothing; the role ids with the fallback action ids
(the static order puts a wait between them and the copies); the columns' copies.  Round 5's library has five."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

ap = argparse.ArgumentParser()
ap.add_argument("kernels", nargs="*", default=["cbh_check_flat_kernel10", "cbh_check_flat_kernel_masks10", "cbh_walk2_kernel10"])
ap.add_argument("--lib", default=os.path.join(ROOT, "cerbos_amd", "libcerbos_hip.so"))
ap.add_argument("--max-bulk-trips", type=int, default=3)
args = ap.parse_args()

with tempfile.TemporaryDirectory() as tmp:
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", args.lib, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                           "--output=" + co, "--unbundle"])
    dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], text=True)

kernels = {}
cur = None
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(_Z\w+)>:", line)
    if m:
        cur = m.group(1)
        kernels[cur] = []
        continue
    if cur is not None and "\t" in line:
        ins = line.split("\t")[1].split("//")[0].strip()
        if ins:
            kernels[cur].append(ins)

bad = 0
for want in args.kernels:
    names = [k for k in kernels if want in k]
    if not names:
        print("%s: no such kernel in %s" % (want, args.lib))
        bad = 1
        continue
    for name in names:
        body = kernels[name]
        end = next((i for i, x in enumerate(body) if x.startswith("s_barrier")), len(body))
        end = min(end, 1200)   # (a kernel without a barrier - one wave to a workgroup: the pre-pass forms - is read that far)
        trips, loads = [], []
        for x in body[:end]:
            op = x.split()[0]
            if op.startswith(("global_load", "buffer_load", "flat_load")):
                loads.append("lds-copy" if "_lds_" in op else op.replace("global_load_", ""))
            elif op == "s_waitcnt" and "vmcnt(0)" in x:
                trips.append(loads)
                loads = []
        carrying = [t for t in trips if t]
        print("%s: %d instructions to the first barrier, %d full waits, %d of them behind loads%s" %
              (name.split("10KernelArgs")[0].lstrip("_Z0123456789"), end, len(trips), len(carrying), " (+ %d loads in flight at the barrier)" % len(loads) if loads else ""))
        for i, t in enumerate(carrying[:16]):
            kinds = {}
            for k in t:
                kinds[k] = kinds.get(k, 0) + 1
            print("    trip %d: %s" % (i + 1, ", ".join("%d x %s" % (n, k) for k, n in sorted(kinds.items()))))
        bulk = [t for t in carrying if sum(1 for k in t if k != "ubyte") >= 8 or "lds-copy" in t]
        if len(bulk) > args.max_bulk_trips:
            print("    %d BULK trips (more than %d): read the disassembly around the extra waits" % (len(bulk), args.max_bulk_trips))
            bad = 1
sys.exit(bad)
