#!/usr/bin/env python3
"""GPU-box probe: cbh_wire_check_pb (the device road in one call) by number of slices - CBH_WIRE_SLICES is read once per process, so
one process per setting:   for s in 1 2 4 8; do CBH_WIRE_SLICES=$s python tools/gpu_wire_onecall.py C2 250000; done"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cerbos_amd import capi, wire, workloads
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

W = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
pol_fn, req_fn = {"C2": (workloads.c2_policies, workloads.c2_requests), "C5": (workloads.c5_policies, workloads.c5_requests),
                  "C3": (workloads.c3_policies, workloads.c3_requests)}[W]
capi.init(0)
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol_fn())))
table = capi.Table(lt.blob)
lib = capi.load()
inputs = req_fn(n).to_inputs()
data, woff = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
pdata = capi.pinned_empty(data.size + 64, np.uint8); pdata[:data.size] = data
cap = 320 * n + 4096
pout, poff, pfl = capi.pinned_empty(cap, np.uint8), capi.pinned_empty(n + 1, np.uint64), capi.pinned_empty(n + 1, np.uint8)
prm = capi.CParams(1_700_000_000_000_000_000, capi.F_WANT_DERIVED_ROLES, 0)
best, tuples = 1e9, 0
for _ in range(10):
    t0 = time.perf_counter()
    info, need = capi.CWireInfo(), C.c_size_t()
    rc = lib.cbh_wire_check_pb(table.h, 0, pdata.ctypes.data, woff.ctypes.data, n, b"default", b"", None, 0, C.byref(prm), pout.ctypes.data, cap,
                               poff.ctypes.data, pfl.ctypes.data, C.byref(need), C.byref(info))
    assert rc == 0, lib.cbh_last_error().decode()
    best, tuples = min(best, time.perf_counter() - t0), info.n_tuples
print("%s n=%d slices=%s: %.3f ms  %.1f M decisions/s  (in %.1f MB, out %.1f MB)" % (W, n, os.environ.get("CBH_WIRE_SLICES", "4"), best * 1e3, tuples / best / 1e6, data.size / 1e6, int(poff[n]) / 1e6))
