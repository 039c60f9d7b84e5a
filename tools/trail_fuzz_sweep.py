#!/usr/bin/env python3
"""One-off wider sweep of AuditTrail.EffectivePolicies on the kernel simulator: generated stores (tests/test_fuzz_parity.py's generator:
scopes, scope permissions, derived roles, role policies with parent roles, principal policies, globs, conditions) x generated
requests, input by input against the oracle, in both scope-search modes - through whatever trail kernel the plan picks for the store
(the walk's twice-walking form, a flat trail kernel, the general walk) and, with CBH_NO_WALK2=1 / CBH_NO_FLAT=1 in the environment,
through the general walk's.
    python tools/trail_fuzz_sweep.py [first_seed] [n_seeds] [requests_per_store]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT]
import numpy as np

import hostsim_api
from cerbos_amd.engine import Conf
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle.check import EvalParams, RuleTableOracle
from test_fuzz_parity import _policies, _requests
from test_hostsim_golden import HostSimEvaluator

first = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
per = int(sys.argv[3]) if len(sys.argv) > 3 else 200
NOW = 1_700_000_000_000_000_000
kinds, compared, bad_total = {}, 0, 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    ev, oracle = HostSimEvaluator(lower_rule_table(rt), Conf()), RuleTableOracle(rt)
    inputs = _requests(rng, per)
    for lenient in (False, True):
        have = ev.effective_policies(inputs, now_ns=NOW, lenient_scope_search=lenient, per_input=True)
        kinds[hostsim_api.last_kind()] = kinds.get(hostsim_api.last_kind(), 0) + 1
        params = EvalParams(now_ns=NOW, lenient_scope_search=lenient)
        want = [oracle.check(i, params)["effectivePolicies"] for i in inputs]
        bad = [k for k in range(len(inputs)) if have[k] != want[k]]
        compared += len(inputs)
        if bad:
            bad_total += len(bad)
            print("seed %d lenient=%s: %d differ, e.g. input %d have %s want %s" % (seed, lenient, len(bad), bad[0], have[bad[0]], want[bad[0]]))
print("stores %d, inputs compared %d, differing %d; kernel family of the runs (0 general, 1 flat, 2 walk): %s" % (count, compared, bad_total, kinds))
sys.exit(1 if bad_total else 0)
