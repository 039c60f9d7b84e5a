"""BASELINE.json configs[0] taken literally: the reference's own load-test sets
(hack/loadtest/templates/{classic,multitenant}, rendered as hack/loadtest/generate.go renders them - fixture
tests/golden/loadtest_templates.json, tools/make_golden_loadtest.py) through every evaluator of the repo.

There is no recorded reference output for these requests (ghz only measures them), so parity here is
oracle/check.py - pinned on the reference's golden engine cases - against the C++ restatement, the kernel
source on the host simulator and (GPU tier) the kernels, in the three evaluation modes.
"""
import numpy as np
import pytest

from cerbos_amd import capi, workloads
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle

NOW = 1_700_000_000_000_000_000
COUNT = 6
MODES = ((False, False), (True, False), (False, True))


def _tables(name, count=COUNT):
    rt = rule_table_from_policies(policies_from_docs(workloads.loadtest_policies(name, count)))
    return rt, lower_rule_table(rt)


def _variants(inputs):
    """The templates' requests plus perturbations that leave the happy path: another tenant's principal, a role the
    policies do not know, an action nobody grants, the parent scope, a missing attribute."""
    out = list(inputs)
    for k, i in enumerate(inputs):
        j = {"requestId": i["requestId"] + "/v", "principal": dict(i["principal"]), "resource": dict(i["resource"]),
             "actions": list(i["actions"]) + ["no_such_action"]}
        m = k % 5
        if m == 0:
            j["principal"]["roles"] = ["stranger"]
        elif m == 1 and j["resource"].get("scope"):
            j["resource"]["scope"] = j["resource"]["scope"].rsplit(".", 1)[0]
        elif m == 2:
            j["principal"]["attr"] = {a: v for a, v in list((j["principal"].get("attr") or {}).items())[1:]}
        elif m == 3:
            j["principal"]["roles"] = list(j["principal"]["roles"]) + ["employee", "admin"]
        else:
            j["resource"]["attr"] = dict(j["resource"].get("attr") or {}, tenantId="tenant_99999", owner=j["principal"]["id"])
        out.append(j)
    return out


def _compare(ev, rt, inputs):
    orc = RuleTableOracle(rt)
    n = 0
    for lenient, strict in MODES:
        outs, bad = ev.check(inputs, now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict, allow_unsupported=True)
        assert not bad, "the load-test sets are inside the device subset"
        params = EvalParams(now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict)
        for inp, have in zip(inputs, outs):
            want = orc.check(inp, params)
            assert norm_actions(have) == norm_actions(want), (lenient, strict, inp)
            assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or []), inp
            n += 1
    return n


@pytest.mark.parametrize("name", ["classic", "multitenant"])
def test_loadtest_set_kernel_source_vs_oracle(name):
    from test_hostsim_golden import HostSimEvaluator
    rt, lt = _tables(name)
    inputs = _variants(workloads.loadtest_inputs(name, COUNT))
    assert _compare(HostSimEvaluator(lt, Conf()), rt, inputs) == 3 * len(inputs)
    # some request of the set must be allowed and some denied, else the comparison says little
    eff = [e["effect"] for i in inputs[:len(inputs) // 2] for e in RuleTableOracle(rt).check(i, EvalParams(now_ns=NOW))["actions"].values()]
    assert "EFFECT_ALLOW" in eff and "EFFECT_DENY" in eff


@pytest.mark.parametrize("name", ["classic", "multitenant"])
def test_loadtest_set_ccheck_vs_kernel_source(name):
    from hostsim_api import check as sim_check
    from oracle import ccheck
    _, lt = _tables(name)
    inputs = _variants(workloads.loadtest_inputs(name, COUNT))
    batch = Flattener(lt).flatten(inputs)
    for flags in (0, capi.F_LENIENT_SCOPE_SEARCH, capi.F_STRICT_EVALUATION):
        flags |= capi.F_WANT_DERIVED_ROLES
        sres = sim_check(lt, batch, NOW, flags)
        try:
            cres = ccheck.check(lt, batch, NOW, flags, 1)
        except ccheck.Unsupported:
            pytest.skip("table outside the C++ restatement")
        ok = cres.status != capi.ST_UNSUPPORTED   # requests whose conditions need general CEL are flagged, not guessed
        for f in ("effect", "policy", "scope"):
            assert np.array_equal(getattr(cres, f)[ok], getattr(sres, f)[ok]), (name, f, flags)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["classic", "multitenant"])
def test_loadtest_set_on_gpu(name):
    rt, lt = _tables(name, 20)
    inputs = _variants(workloads.loadtest_inputs(name, 20))
    ev = HipEvaluator(lt, Conf())
    try:
        assert _compare(ev, rt, inputs) == 3 * len(inputs)
    finally:
        ev.close()
