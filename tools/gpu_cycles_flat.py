#!/usr/bin/env python3
"""GPU-box profiling aid for cbh_check_flat_kernel (library built with CBH_PROFILE=1): per-wave phase cycles, wall-clock
start / end of every wave (dispatch ramp, drain, lifetime) and record / round counts.   python tools/gpu_cycles_flat.py C2|C3|C4"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cerbos_amd import capi, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

W = sys.argv[1] if len(sys.argv) > 1 else "C2"
pol_fn, req_fn, n = {"C2": (workloads.c2_policies, workloads.c2_requests, 250_000),
                     "C3": (workloads.c3_policies, workloads.c3_requests, 1_000_000),
                     "C4": (workloads.c4_policies, workloads.c4_requests, 500_000),
                     "T": (lambda: workloads.c4_policies(seed=7, n_policies=100, rules_per_policy=100),
                           lambda n: workloads.c4_requests(n, seed=7, n_policies=100), 250_000)}[W]
capi.init(0)
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol_fn())))
table = capi.Table(lt.blob)
batch = req_fn(n).to_batch(Flattener(lt))
db = table.upload(batch)
for _ in range(3):
    table.launch(db, now_ns=1, flags=0x100)
table.synchronize()
res = table.download(db)
assert (batch.req_u32[9] == 4).all(), "the tool reads lane 0's four policy / scope words"
pol = res.policy.reshape(-1, 4)[::64].astype(np.int64)
scp = res.scope.reshape(-1, 4)[::64].astype(np.int64)
print(W, "waves", len(pol))
for name, a in (("loads+classes", pol[:, 0]), ("eval (in walk)", pol[:, 1]), ("walk", pol[:, 2]), ("staging (in walk)", pol[:, 3]),
                ("records staged", scp[:, 2] & 0xFFFF), ("evals", scp[:, 2] >> 16), ("visits", scp[:, 3] >> 16), ("rounds", scp[:, 3] & 0xFFFF)):
    print("%-14s min %8d  p10 %8d  p50 %8d  p90 %8d  max %8d  mean %10.1f" % (name, a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max(), a.mean()))
t0 = (scp[:, 0] - scp[:, 0].min()) & 0xFFFFFFFF
t1 = (scp[:, 1] - scp[:, 0].min()) & 0xFFFFFFFF
life = (t1 - t0) / 100.0   # us (100 MHz)
print("wave start  us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(t0 / 100.0, [10, 50, 90, 100])))
print("wave end    us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(t1 / 100.0, [10, 50, 90, 100])))
print("wave life   us: p10 %.2f p50 %.2f p90 %.2f max %.2f mean %.2f" % (*np.percentile(life, [10, 50, 90, 100]), life.mean()))
order = np.argsort(t0)
nb = 10
for k in range(nb):   # by position in the grid: when did these waves start / end
    sl = slice(k * len(pol) // nb, (k + 1) * len(pol) // nb)
    print("  grid decile %d: start %.2f  end %.2f  life %.2f  records %.0f  rounds %.1f" % (k, t0[sl].mean() / 100.0, t1[sl].mean() / 100.0, life[sl].mean(), (scp[sl, 2] & 0xFFFF).mean(), (scp[sl, 3] & 0xFFFF).mean()))
for _ in range(20):
    table.launch(db, now_ns=1, flags=0)
table.synchronize()
print("kernel ms", table.kernel_time_ms())
