#!/usr/bin/env python3
"""One-off: corrupted CheckResourcesRequests (truncations, bit flips, inserted bytes) down the whole request road of the LIBRARY's
simulator build - run it under AddressSanitizer (tools/sim_engine_asan.sh builds the library; see the command below): "device" memory
is heap memory there, so any kernel of the road (count, split, flattener, decision, assembler) that reads or writes outside an
allocation on a malformed message aborts.  A call either names a malformed request / leaves it to the host flattener, or answers; the
good requests around the corrupted one must then get the answers they get alone.
    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \\
      CBH_TEST_SIM_LIB=/tmp/cbh_sim_asan/libcerbos_hip_sim_asan.so python tools/request_road_corruption_sweep.py [trials]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT]
import numpy as np

from sim_engine import sim_engine
from cerbos_amd import wire
from cerbos_amd.lower.blob import lower_rule_table
from helpers import load_json, store_rule_table

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
NOW = 1_700_000_000_000_000_000
GLOBALS = {"environment": "test"}
cases = [c["inputs"] for c in load_json("server_check_cases.json")] + [c["inputs"] for c in load_json("engine_cases.json") if not c["wantError"]][:20]
cases = [g for g in cases if not any("auxData" in i for i in g)]


def request_of(g, meta):
    return {"requestId": g[0].get("requestId", ""), "includeMeta": meta, "principal": g[0]["principal"],
            "resources": [{"actions": i["actions"], "resource": i["resource"]} for i in g]}


refused = host = answered = 0
with sim_engine() as capi:
    table = capi.Table(lower_rule_table(store_rule_table(), GLOBALS).blob)
    rng = np.random.default_rng(17)
    alone = {}
    for trial in range(trials):
        g = cases[trial % len(cases)]
        m = bytearray(wire.encode_check_resources_request(request_of(g, bool(trial & 1))))
        pos = int(rng.integers(0, len(m)))
        if trial % 3 == 0:
            del m[pos:]
        elif trial % 3 == 1:
            m[pos] ^= 1 << int(rng.integers(0, 8))
        else:
            m[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 5)), dtype=np.uint8))
        k = (trial + 1) % len(cases)
        good = wire.encode_check_resources_request(request_of(cases[k], False))
        if k not in alone:
            alone[k] = table.wire_check_requests_pb([good], now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES, trail=True)
        try:
            outs, flags, meta, masks = table.wire_check_requests_pb([good, bytes(m), good], now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES, trail=True)
        except capi.HostFlattenerNeeded:
            host += 1
            continue
        except capi.HipEngineError as e:
            assert "malformed" in str(e) or "too large" in str(e), e
            refused += 1
            continue
        answered += 1
        assert outs[0] == alone[k][0][0] and outs[2] == alone[k][0][0], trial
        assert np.array_equal(masks[0], alone[k][3][0]) and np.array_equal(masks[2], alone[k][3][0]), trial
    table.close()
print("corrupted requests: %d refused by name, %d left to the host flattener, %d answered (their good neighbours' outputs and trails untouched)"
      % (refused, host, answered))
