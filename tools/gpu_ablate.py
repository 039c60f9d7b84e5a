#!/usr/bin/env python3
"""GPU-box measurement aid: decision-kernel time with parts of the walk switched off.

Needs a library built with CBH_ABLATION=1 (tools/build_all.sh); in a production build the flags
are ignored and every line prints the same time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerbos_amd import capi, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

capi.init(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
pol, reqs, n = {"C2": (workloads.c2_policies, workloads.c2_requests, 250_000),
                "C3": (workloads.c3_policies, workloads.c3_requests, 1_000_000)}[wl]
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
table = capi.Table(lt.blob)
batch = reqs(n).to_batch(Flattener(lt))
print("workload", wl, "requests", n)
db = table.upload(batch)
VARIANTS = (("full", 0), ("no_eval", 0x400), ("no_rows", 0x800), ("no_passes", 0x200))
if len(sys.argv) > 2:   # one variant only (for counter collection: tools/gpu_pmc_ablate.sh)
    VARIANTS = tuple(v for v in VARIANTS if v[1] == int(sys.argv[2], 0))
BASE = 4   # CBH_F_WANT_DERIVED_ROLES, as bench.py
for name, fl in VARIANTS:
    fl |= BASE
    for _ in range(5):
        table.launch(db, now_ns=1, flags=fl)
    table.synchronize()
    table.kernel_time_ms()
    for _ in range(30):
        table.launch(db, now_ns=1, flags=fl)
    table.synchronize()
    ck, _ = table.kernel_time_ms()
    print("ablation %-10s kernel %.4f ms" % (name, ck))
