"""The reference's engine-case inputs, mutated as an unvalidated caller of ``engine.Check`` could hand them over
(empty principal id, zero / duplicate / many roles and actions, unknown kinds and versions, malformed scopes, attributes
retyped / nested / dropped, non-ASCII strings), under five evaluation modes, through the dict path and the bytes-in /
bytes-out device road, against ``oracle/check.py`` (tests/mutation_probe.py).  CPU tier: the kernel source on the host
simulator; GPU tier: the library on the MI355X."""
import pytest

import mutation_probe as mp
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from helpers import store_rule_table

GLOBALS = {"environment": "test"}


def _assert_clean(r, min_probes):
    assert not r["wrong"], "%d of %d probes differ from the oracle, e.g. %s" % (len(r["wrong"]), r["probes"], r["wrong"][:3])
    assert r["probes"] >= min_probes
    assert r["flagged"] * 50 < r["probes"]      # the device path answers the mutated inputs itself


def test_mutated_inputs_on_the_simulator():
    from test_hostsim_golden import HostSimEvaluator
    rt = store_rule_table()
    ev = HostSimEvaluator(lower_rule_table(rt, GLOBALS), Conf(globals_=GLOBALS))
    _assert_clean(mp.run_probe(ev, rt, GLOBALS, mp.reference_inputs()), 50_000)


def test_empty_principal_id_takes_every_principal_policy():
    """index/index.go:228-234: no principal filter when the id is ""; check.go names the policy after the (empty) id."""
    from test_hostsim_golden import HostSimEvaluator
    rt = store_rule_table()
    ev = HostSimEvaluator(lower_rule_table(rt, GLOBALS), Conf(globals_=GLOBALS))
    inp = {"requestId": "t", "actions": ["approve", "view:public"],
           "principal": {"id": "", "policyVersion": "20210210", "roles": ["employee"], "attr": {"department": "marketing", "geography": "GB", "team": "design", "managed_geographies": "GB"}},
           "resource": {"kind": "leave_request", "policyVersion": "20210210", "id": "XX125", "attr": {"department": "marketing", "geography": "GB", "id": "XX125", "owner": "john", "team": "design", "status": "PENDING_APPROVAL", "dev_record": True}}}
    from oracle.check import EvalParams, RuleTableOracle
    want = RuleTableOracle(rt).check(inp, EvalParams(globals_=GLOBALS, now_ns=mp.NOW))
    assert any(e["policy"].startswith("principal..v") for e in want["actions"].values())   # the oracle does take principal rows
    outs, bad = ev.check([inp], now_ns=mp.NOW, allow_unsupported=True)
    assert not bad and mp.norm_actions(outs[0]) == mp.norm_actions(want)


@pytest.mark.gpu
def test_mutated_inputs_on_the_gpu():
    rt = store_rule_table()
    ev = HipEvaluator(lower_rule_table(rt, GLOBALS), Conf(globals_=GLOBALS))
    try:
        _assert_clean(mp.run_probe(ev, rt, GLOBALS, mp.reference_inputs()), 50_000)
    finally:
        ev.close()
