#!/usr/bin/env python3
"""GPU-box probe of the one-shot path variants (CBH_TRACE lines on stderr): run as
   CBH_TRACE=1 [CBH_COPY_MODE=1] [CBH_CHUNK_REQUESTS=n] [CBH_ZEROCOPY_BYTES=n] [CBH_SPIN=1] python tools/gpu_probe_oneshot.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerbos_amd import capi, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
import ctypes as C

lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c2_policies())))
capi.init(0)
table = capi.Table(lt.blob)
fl = Flattener(lt)
cr = workloads.c2_requests(250_000)
NOW = 1_700_000_000_000_000_000
lib = capi.load()
big = capi.pin_batch(cr.to_batch(fl))
for want, tag in (((), "effect-only"), (("policy", "scope", "status", "edr"), "all-outputs")):
    into = capi.Result(big.n_tuples, big.n_requests, want, True)
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter(); table.check(big, now_ns=NOW, flags=4, want=want, device_order=True, into=into); best = min(best, time.perf_counter() - t0)
    print("big pinned %s: %.3f ms  %.3f G decisions/s" % (tag, best * 1e3, big.n_tuples / best / 1e9), flush=True)
for n in (12, 100, 1000):
    small = fl.flatten(cr.to_inputs(0, n))
    cb, prm = capi.make_cbatch(small, table.num_columns), capi.CParams(NOW, 4, 0)
    res = capi.Result(small.n_tuples, small.n_requests, ("policy", "scope", "status", "edr"))
    lat = []
    for _ in range(300):
        t0 = time.perf_counter(); rc = lib.cbh_check_batch(table.h, C.byref(cb), C.byref(prm), C.byref(res.c)); lat.append(time.perf_counter() - t0)
    print("small n=%d requests: p50 %.1f us  p10 %.1f us" % (n, np.median(lat[50:]) * 1e6, np.percentile(lat[50:], 10) * 1e6), flush=True)
