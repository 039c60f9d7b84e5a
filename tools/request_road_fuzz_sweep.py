#!/usr/bin/env python3
"""One-off sweep of the request road on the simulator build of the LIBRARY (tests/sim_engine.py): generated stores x generated
requests, grouped at random into CheckResourcesRequests (0-12 resource entries; the first input's principal, as svc.CheckResources
builds them) - cbh_wire_check_requests_trail_pb against cbh_wire_check_pb on the CheckInputs those requests stand for (byte for byte the
same CheckOutputs and flags) and every request's trail against the oracle's union over its entries.
    python tools/request_road_fuzz_sweep.py [first_seed] [n_seeds] [inputs_per_store]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT]
import numpy as np

from sim_engine import sim_engine
from cerbos_amd import wire
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle.check import EvalParams, RuleTableOracle
from test_fuzz_parity import _policies, _requests

first = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
per = int(sys.argv[3]) if len(sys.argv) > 3 else 300
NOW = 1_700_000_000_000_000_000
n_req = n_in = host = trail_checked = 0
with sim_engine() as capi:
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
        lt = lower_rule_table(rt)
        ev, oracle = HipEvaluator(lt, Conf()), RuleTableOracle(rt)
        inputs = [i for i in _requests(rng, per) if len(i["actions"]) <= 64]   # (more actions: the host flattener's, by design)
        groups, k = [], 0
        while k < len(inputs):
            m = int(rng.integers(0, 13))
            groups.append(inputs[k:k + m])
            k += m
        built = [[dict({kk: v for kk, v in i.items() if kk != "auxData"}, principal=g[0]["principal"], requestId=g[0].get("requestId", "")) for i in g] for g in groups]
        reqs = [wire.encode_check_resources_request({"requestId": g[0].get("requestId", ""), "principal": g[0]["principal"],
                                                     "resources": [{"actions": i["actions"], "resource": i["resource"]} for i in g]}) if g else b"" for g in groups]
        try:
            outs, flags, _, trails = ev.check_requests_pb(reqs, now_ns=NOW, audit_trail=True)
        except capi.HostFlattenerNeeded as e:
            host += 1
            if host <= 2:
                print("seed %d: %s" % (seed, e))
            ev.close()
            continue
        flat = [i for g in built for i in g]
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in flat])
        want, want_flags = ev.table.wire_check_pb(data, off, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        assert [b for o in outs for b in o] == want and np.array_equal(flags, want_flags), seed
        params = EvalParams(now_ns=NOW)
        at = 0
        for g, trail in zip(built, trails):
            flagged = any(flags[at + j] & 1 for j in range(len(g)))
            at += len(g)
            if flagged:
                continue
            keys = set()
            for i in g:
                keys.update(oracle.check(i, params)["effectivePolicies"])
            assert trail == sorted(keys), (seed, g)
            trail_checked += 1
        n_req += len(groups); n_in += len(flat)
        ev.close()
print("stores %d (%d left to the host flattener), requests %d, resource entries %d: same bytes by both roads; %d request trails equal to the oracle's"
      % (count, host, n_req, n_in, trail_checked))
