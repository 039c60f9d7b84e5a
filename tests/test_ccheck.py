"""Pins oracle/ccheck.cpp (the scalar C++ restatement used for full-size parity and as the CPU
baseline) against oracle/check.py, which is itself pinned on the reference's golden fixtures
(tests/test_oracle_golden.py):

* the synthetic BASELINE configurations in every evaluation mode - effect, policy key, scope,
  effective derived roles of every tuple, and whether the request produced evaluation errors;
* the reference's own golden store with its role policies taken out (ccheck reports tables with
  role policies / parent roles as unsupported) over the inputs of all golden engine cases.
"""
import json
import os

import numpy as np
import pytest

from cerbos_amd import capi, workloads
from cerbos_amd.engine import HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle import ccheck
from oracle.check import EvalParams, RuleTableOracle

NOW = 1_700_000_000_000_000_000
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CONFIGS = {
    "C1": (lambda: workloads.c1_policies(2), lambda n: workloads.c1_requests(n, n_sets=2)),
    "C2": (workloads.c2_policies, lambda n: workloads.c2_requests(n)),
    "C3": (workloads.c3_policies, lambda n: workloads.c3_requests(n)),
}


class _Decoder(HipEvaluator):
    """ids -> CheckOutput through the product's own assembly code, without a GPU table."""

    def __init__(self, lt):   # noqa: D401 - deliberately skips HipEvaluator.__init__ (needs a device)
        self.lt = lt


def _compare(rt, lt, inputs, batch, mode, threads=1, check_errors=True):
    flags = capi.F_WANT_DERIVED_ROLES
    flags |= capi.F_LENIENT_SCOPE_SEARCH if mode == "lenient" else 0
    flags |= capi.F_STRICT_EVALUATION if mode == "strict" else 0
    res = ccheck.check(lt, batch, NOW, flags, threads)
    outs, bad = _Decoder(lt).assemble(inputs, batch, res, "default", allow_unsupported=True)
    orc = RuleTableOracle(rt)
    params = EvalParams(now_ns=NOW, lenient_scope_search=mode == "lenient", strict_evaluation=mode == "strict")
    t = 0
    n_cmp = 0
    for r, (inp, got) in enumerate(zip(inputs, outs)):
        n_act = len(batch.actions_per_request[r])
        st = res.status[t:t + n_act]
        t += n_act
        if r in bad:
            continue
        want = orc.check(inp, params)
        for a, w in want["actions"].items():
            assert got["actions"][a] == {"effect": w["effect"], "policy": w["policy"], "scope": w.get("scope", "")}, (r, a, inp)
        assert sorted(got["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles", [])), (r, inp)
        if check_errors:
            assert bool((st == capi.ST_CEL_ERROR).any()) == bool(want.get("evaluationErrors")), (r, inp, want.get("evaluationErrors"))
        n_cmp += 1
    return n_cmp, len(bad)


@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("mode", ["default", "lenient", "strict"])
def test_ccheck_matches_python_oracle_on_configs(name, mode):
    pol_fn, req_fn = CONFIGS[name]
    rt = rule_table_from_policies(policies_from_docs(pol_fn()))
    lt = lower_rule_table(rt)
    cr = req_fn(500)
    inputs = cr.to_inputs()
    n_cmp, n_bad = _compare(rt, lt, inputs, Flattener(lt).flatten(inputs), mode)
    assert n_bad == 0 and n_cmp == len(inputs)


def test_ccheck_threads_agree():
    rt = rule_table_from_policies(policies_from_docs(workloads.c3_policies()))
    lt = lower_rule_table(rt)
    batch = workloads.c3_requests(20_000).to_batch(Flattener(lt))
    a = ccheck.check(lt, batch, NOW, 0, threads=1)
    b = ccheck.check(lt, batch, NOW, 0, threads=4)
    for f in ("effect", "policy", "scope", "status", "edr"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_ccheck_on_golden_store_without_role_policies():
    with open(os.path.join(GOLDEN, "store_policies.json")) as fh:
        docs = [d for d in json.load(fh) if "rolePolicy" not in d]
    with open(os.path.join(GOLDEN, "engine_cases.json")) as fh:
        cases = json.load(fh)
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    inputs = [inp for c in cases for inp in c["inputs"]]
    assert len(inputs) > 50
    total = skipped = 0
    for mode in ("default", "lenient", "strict"):
        # error presence is not compared here: the reference evaluates a policy's variables eagerly and
        # records their errors even when no condition reads them; the lowering inlines variables
        n_cmp, n_bad = _compare(rt, lt, inputs, Flattener(lt).flatten(inputs, sort=False), mode, check_errors=False)
        total += n_cmp
        skipped += n_bad
    # requests whose policies use general CEL programs are reported unsupported, never guessed
    assert total > skipped, (total, skipped)


def test_ccheck_refuses_role_policy_tables():
    with open(os.path.join(GOLDEN, "store_policies.json")) as fh:
        docs = json.load(fh)
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    batch = Flattener(lt).flatten([{"principal": {"id": "x", "roles": ["user"]}, "resource": {"kind": "leave_request", "id": "1"},
                                    "actions": ["view"]}])
    with pytest.raises(ccheck.Unsupported):
        ccheck.check(lt, batch, NOW, 0)


@pytest.mark.parametrize("name", ["C2", "C3"])
@pytest.mark.parametrize("mode", ["default", "lenient", "strict"])
def test_device_source_bit_exact_against_ccheck(name, mode):
    """The decision kernel's source (host-simulated waves) vs the C++ restatement on 3000 requests:
    every output array identical, and the same requests report CEL errors."""
    import hostsim_api
    pol_fn = CONFIGS[name][0]
    full = {"C2": workloads.c2_requests, "C3": workloads.c3_requests}[name]
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol_fn())))
    batch = full(3000).to_batch(Flattener(lt))
    flags = capi.F_WANT_DERIVED_ROLES
    flags |= capi.F_LENIENT_SCOPE_SEARCH if mode == "lenient" else 0
    flags |= capi.F_STRICT_EVALUATION if mode == "strict" else 0
    got = hostsim_api.check(lt, batch, NOW, flags)
    want = ccheck.check(lt, batch, NOW, flags)
    for f in ("effect", "policy", "scope", "edr"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), f
    ge = (got.status == capi.ST_CEL_ERROR).reshape(-1, 4).any(axis=1)
    we = (want.status == capi.ST_CEL_ERROR).reshape(-1, 4).any(axis=1)
    assert np.array_equal(ge, we)
