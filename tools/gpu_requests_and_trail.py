#!/usr/bin/env python3
"""GPU-box probe for the two roads written after round 4's GPU minutes were spent:
  * cbh_wire_check_requests_pb (CheckResourcesRequests down the device road) against cbh_wire_check_pb on the CheckInputs those requests
    stand for: same answers (checked), decisions/s of either from one caller thread;
  * cbh_check_batch_trail (the general walk + the effective-policy masks) against cbh_check_batch on the same batch.
     python tools/gpu_requests_and_trail.py C2 250000 10      (workload, inputs, resource entries per request)"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cerbos_amd import capi, wire, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

W = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
per = int(sys.argv[3]) if len(sys.argv) > 3 else 10
pol_fn, req_fn = {"C2": (workloads.c2_policies, workloads.c2_requests), "C5": (workloads.c5_policies, workloads.c5_requests),
                  "C3": (workloads.c3_policies, workloads.c3_requests)}[W]
NOW = 1_700_000_000_000_000_000
capi.init(0)
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol_fn())))
table = capi.Table(lt.blob)
lib = capi.load()
inputs = req_fn(n).to_inputs()
groups = [inputs[k:k + per] for k in range(0, n, per)]
service = [dict({k: v for k, v in i.items() if k != "auxData"}, principal=g[0]["principal"], requestId=g[0].get("requestId", "")) for g in groups for i in g]
reqs = [wire.encode_check_resources_request({"requestId": g[0].get("requestId", ""), "principal": g[0]["principal"],
                                             "resources": [{"actions": i["actions"], "resource": i["resource"]} for i in g]}) for g in groups]
rdata, roff = wire.pack_messages(reqs)
idata, ioff = wire.pack_messages([wire.encode_check_input(i) for i in service])
prm = capi.CParams(NOW, capi.F_WANT_DERIVED_ROLES, 0)
cap = 320 * n + 4096


def pinned(a):
    p = capi.pinned_empty(a.size + 64, np.uint8)
    p[:a.size] = a
    return p


prd, pid = pinned(rdata), pinned(idata)
pout, poff, pfl = capi.pinned_empty(cap, np.uint8), capi.pinned_empty(n + 1, np.uint64), capi.pinned_empty(n + 1, np.uint8)
first = np.zeros(len(reqs) + 1, np.uint32)
rflags = np.zeros(len(reqs) + 1, np.uint8)
answers = {}
ep = np.zeros((len(reqs) + 1, (len(lt.policy_keys) + 31) // 32 or 1), np.uint32)
lib.cbh_wire_check_requests_trail_pb.argtypes = lib.cbh_wire_check_requests_pb.argtypes + [C.c_void_p]
for name in ("inputs", "requests", "requests + trail"):
    best, tuples = 1e9, 0
    for _ in range(8):
        info, need = capi.CWireInfo(), C.c_size_t()
        t0 = time.perf_counter()
        if name == "inputs":
            rc = lib.cbh_wire_check_pb(table.h, 0, pid.ctypes.data, ioff.ctypes.data, n, b"default", b"", None, 0, C.byref(prm), pout.ctypes.data, cap,
                                       poff.ctypes.data, pfl.ctypes.data, C.byref(need), C.byref(info))
        elif name == "requests":
            rc = lib.cbh_wire_check_requests_pb(table.h, 0, prd.ctypes.data, roff.ctypes.data, len(reqs), None, None, b"default", b"", None, 0, C.byref(prm),
                                                first.ctypes.data, rflags.ctypes.data, pout.ctypes.data, cap, poff.ctypes.data, pfl.ctypes.data, n,
                                                C.byref(need), C.byref(info))
        else:   # the audit trail of every request beside its outputs (flat tables: the flat trail kernels; others: the general walk)
            rc = lib.cbh_wire_check_requests_trail_pb(table.h, 0, prd.ctypes.data, roff.ctypes.data, len(reqs), None, None, b"default", b"", None, 0, C.byref(prm),
                                                      first.ctypes.data, rflags.ctypes.data, pout.ctypes.data, cap, poff.ctypes.data, pfl.ctypes.data, n,
                                                      C.byref(need), C.byref(info), ep.ctypes.data)
        dt = time.perf_counter() - t0
        assert rc == 0, lib.cbh_last_error().decode()
        best, tuples = min(best, dt), info.n_tuples
    answers[name] = bytes(pout[:int(poff[n])])
    print("%s %s n=%d (%d per request): %.3f ms  %.1f M decisions/s  (in %.1f MB)" % (W, name, n, per, best * 1e3, tuples / best / 1e6, (idata if name == "inputs" else rdata).size / 1e6))
assert answers["inputs"] == answers["requests"] == answers["requests + trail"], "the roads disagree"
print("same %d bytes of CheckOutputs by both roads" % len(answers["inputs"]))

# ---- the trail
m = min(n, 200_000)
batch = Flattener(lt).flatten(inputs[:m], "default", "")
for name in ("check_batch", "check_batch_trail"):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        if name == "check_batch":
            res = table.check(batch, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES, device_order=True)
        else:
            res2, masks = table.check_trail(batch, None, 1, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        best = min(best, time.perf_counter() - t0)
    print("%s %s: %.3f ms for %d tuples  %.1f M decisions/s (PCIe-inclusive, pageable arrays)" % (W, name, best * 1e3, batch.n_tuples, batch.n_tuples / best / 1e6))
assert np.array_equal(res.effect, res2.effect) and np.array_equal(res.policy, res2.policy)
from cerbos_amd.engine import effective_policy_keys
keys = effective_policy_keys(lt.policy_keys, masks[0])
print("effective policies of the batch: %d, e.g. %s" % (len(keys), keys[:4]))
