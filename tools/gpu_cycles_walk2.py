#!/usr/bin/env python3
"""GPU-box profiling aid for cbh_walk2_kernel / cbh_walk2_pre_kernel (library built with -DCBH_PROFILE_CYCLES): per-wave phase
cycles, wave lifetimes, visit counts.   python tools/gpu_cycles_walk2.py C5        (CBH_PRE_ONLY=1: the pre-pass's figures)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cerbos_amd import capi, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

W = sys.argv[1] if len(sys.argv) > 1 else "C5"
pol_fn, req_fn, n = {"C2": (workloads.c2_policies, workloads.c2_requests, 250_000),
                     "C3": (workloads.c3_policies, workloads.c3_requests, 1_000_000),
                     "C4": (workloads.c4_policies, workloads.c4_requests, 500_000),
                     "C5": (workloads.c5_policies, workloads.c5_requests, 250_000)}[W]
pre = os.environ.get("CBH_PRE_ONLY") is not None
capi.init(0)
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol_fn())))
table = capi.Table(lt.blob)
batch = req_fn(n).to_batch(Flattener(lt))
db = table.upload(batch)
for _ in range(3):
    table.launch(db, now_ns=1, flags=0x100 | 4)
table.synchronize()
res = table.download(db)
assert (batch.req_u32[9] == 4).all()
pol = res.policy.reshape(-1, 4)[::64].astype(np.int64)
scp = res.scope.reshape(-1, 4)[::64].astype(np.int64)
print(W, "pre-pass" if pre else "walk", "waves", len(pol))
if os.environ.get("CBH_PROFILE_PARTS") and not pre:   # library built with -DCBH_PROFILE_CYCLES=2: the prologue and the principal pass in parts
    for name, a in (("request fields", pol[:, 0]), ("column cache", pol[:, 1]), ("ids classes globs", pol[:, 2]), ("parent roles", pol[:, 3]),
                    ("sets, wave ORs", scp[:, 2] & 0xFFFFF), ("principal: who", (scp[:, 2] >> 20) << 4), ("principal: walk", scp[:, 3])):
        print("%-18s min %8d  p10 %8d  p50 %8d  p90 %8d  max %8d  mean %10.1f" % (name, a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max(), a.mean()))
    sys.exit(0)
names = ("prologue", "principal", "walk", "eval_sum" if pre else "fold+edr")
for name, a in list(zip(names, (pol[:, 0], pol[:, 1], pol[:, 2], pol[:, 3]))) + [("records", scp[:, 2] & 0xFFFF), ("evals", scp[:, 2] >> 16), ("rounds", scp[:, 3])]:
    print("%-14s min %8d  p10 %8d  p50 %8d  p90 %8d  max %8d  mean %10.1f" % (name, a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max(), a.mean()))
t0 = (scp[:, 0] - scp[:, 0].min()) & 0xFFFFFFFF
t1 = (scp[:, 1] - scp[:, 0].min()) & 0xFFFFFFFF
life = (t1 - t0) / 100.0
print("wave start  us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(t0 / 100.0, [10, 50, 90, 100])))
print("wave end    us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(t1 / 100.0, [10, 50, 90, 100])))
print("wave life   us: p10 %.2f p50 %.2f p90 %.2f max %.2f mean %.2f" % (*np.percentile(life, [10, 50, 90, 100]), life.mean()))
for _ in range(20):
    table.launch(db, now_ns=1, flags=4)
table.synchronize()
print("plan ms", table.kernel_time_ms())
