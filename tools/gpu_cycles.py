#!/usr/bin/env python3
"""GPU-box profiling aid: per-wave cycle counts of the decision kernel (CBH_F_DEBUG_CYCLES)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cerbos_amd import capi, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

W = sys.argv[1] if len(sys.argv) > 1 else "C2"
pol_fn, req_fn, nreq = {"C2": (workloads.c2_policies, workloads.c2_requests, 250_000),
                        "C3": (workloads.c3_policies, workloads.c3_requests, 1_000_000),
                        "C5": (workloads.c5_policies, workloads.c5_requests, 250_000)}[W]
capi.init(0)
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol_fn())))
table = capi.Table(lt.blob)
cr = req_fn(nreq)
batch = cr.to_batch(Flattener(lt))
db = table.upload(batch)
for _ in range(3):
    table.launch(db, now_ns=1, flags=0x100)
table.synchronize()
res = table.download(db)
pol = res.policy.reshape(-1, 4)[::64]          # lane 0 of every wave
pre, body, ev, nev = (pol[:, k].astype(np.int64) for k in range(4))
print("waves", len(pre))
scp = res.scope.reshape(-1, 4)[::64]
ra, rb, rc, rd = (scp[:, k].astype(np.int64) for k in range(4))
for name, a in (("preamble", pre), ("passes", body), ("eval_sum", ev), ("n_evals", nev), ("role_setup", ra),
                ("bucket_dir", rb), ("row_match", rc), ("row_post", rd)):
    print("%-9s min %8d  p50 %8d  p90 %8d  max %8d" % (name, a.min(), np.median(a), np.percentile(a, 90), a.max()))

# ablations (profile build only): kernel time with parts of the walk switched off
for name, fl in (("full", 0), ("no_eval", 0x400), ("no_rows", 0x800), ("no_passes", 0x200)):
    for _ in range(20):
        table.launch(db, now_ns=1, flags=fl)
    table.synchronize()
    ck, _ = table.kernel_time_ms()
    print("ablation %-10s kernel %.4f ms" % (name, ck))
