"""What the device table has no room for is lowered LOUDLY and per rule, not by refusing the table (round 3 raised
LoweringError for a 65th glob pattern or derived role name - the whole table fell back to the caller's engine):
* a glob pattern beyond the 64 match bits / 512 automaton positions of its dimension is lowered as "*" with a condition
  that flags whoever evaluates it;
* a derived role beyond the 64 bits of the effective-derived-roles mask keeps its definition under a condition that flags
  the requests whose roles reach it.
Every request the device decides must agree with oracle/check.py; every request it cannot decide faithfully must come
back CBH_ST_UNSUPPORTED - never a wrong effect.  (The reference has neither limit: index/glob_dimension.go:31-118.)"""
import numpy as np
import pytest

import hostsim_api
from cerbos_amd import capi
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000


def _compare(docs, inputs, run=None):
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    batch = Flattener(lt).flatten(inputs)
    res = (run or (lambda lt_, b: hostsim_api.check(lt_, b, NOW, capi.F_WANT_DERIVED_ROLES)))(lt, batch)
    orc = RuleTableOracle(rt)
    decided = flagged = t = 0
    for inp in inputs:
        want = orc.check(inp, EvalParams(now_ns=NOW))
        na = len(inp["actions"])
        st = res.status[t:t + na]
        if (st == capi.ST_UNSUPPORTED).any():
            flagged += 1
        else:
            for k, a in enumerate(inp["actions"]):
                assert (res.effect[t + k] == capi.EFFECT_ALLOW) == (want["actions"][a]["effect"] == "EFFECT_ALLOW"), (inp, a)
            decided += 1
        t += na
    return lt, decided, flagged


def _glob_store(n_patterns):
    rules = [{"actions": ["view"], "roles": ["user"], "effect": "EFFECT_ALLOW"}]
    for i in range(n_patterns):
        rules.append({"actions": ["grp%02d:*" % i], "roles": ["user"], "effect": "EFFECT_ALLOW" if i % 3 else "EFFECT_DENY"})
    return [{"apiVersion": API, "resourcePolicy": {"resource": "doc", "version": "default", "rules": rules}}]


def _glob_inputs(n_patterns):
    out = []
    for i in range(n_patterns):
        out.append({"requestId": "g%d" % i, "actions": ["grp%02d:read" % i, "view"], "principal": {"id": "p", "roles": ["user"]},
                    "resource": {"kind": "doc", "id": "d%d" % i}})
    out.append({"requestId": "lit", "actions": ["view"], "principal": {"id": "p", "roles": ["user"]}, "resource": {"kind": "doc", "id": "x"}})
    out.append({"requestId": "none", "actions": ["nothing:here"], "principal": {"id": "p", "roles": ["stranger"]}, "resource": {"kind": "doc", "id": "y"}})
    return out


def test_a_65th_glob_pattern_flags_its_rule_not_the_table():
    lt, decided, flagged = _compare(_glob_store(80), _glob_inputs(80))
    assert any("glob patterns" in reason for _e, reason in lt.unsupported)
    # a pattern lowered as "*" matches every action: whoever meets one of those rules with a role that matches is flagged -
    # here every request of role `user` (all of them reach the over-matching rules); the stranger is decided
    assert decided >= 1 and flagged >= 16
    # within the limits nothing is flagged
    lt, decided, flagged = _compare(_glob_store(60), _glob_inputs(60))
    assert flagged == 0 and decided == 62 and not lt.unsupported


def _dr_store(n_names):
    defs = [{"name": "dr%02d" % i, "parentRoles": ["role%02d" % i], "condition": {"match": {"expr": "R.attr.owner == P.id"}}} for i in range(n_names)]
    rules = [{"actions": ["view"], "roles": ["user"], "effect": "EFFECT_ALLOW"}]
    rules += [{"actions": ["edit"], "derivedRoles": ["dr%02d" % i], "effect": "EFFECT_ALLOW"} for i in range(n_names)]
    return [{"apiVersion": API, "derivedRoles": {"name": "many", "definitions": defs}},
            {"apiVersion": API, "resourcePolicy": {"resource": "doc", "version": "default", "importDerivedRoles": ["many"], "rules": rules}}]


def _dr_inputs(n_names):
    out = []
    for i in range(n_names):
        out.append({"requestId": "q%d" % i, "actions": ["edit", "view"], "principal": {"id": "p%d" % (i % 2), "roles": ["role%02d" % i, "user"]},
                    "resource": {"kind": "doc", "id": "d", "attr": {"owner": "p0"}}})
    return out


def test_a_65th_derived_role_flags_the_requests_it_concerns():
    lt, decided, flagged = _compare(_dr_store(70), _dr_inputs(70))
    assert len(lt.dr_names) == 64
    assert decided == 64 and flagged == 6      # the requests whose role is a parent of one of the six names without a bit
    lt, decided, flagged = _compare(_dr_store(64), _dr_inputs(64))
    assert flagged == 0 and decided == 64


@pytest.mark.gpu
def test_capacity_limits_on_the_gpu():
    tables = []

    def run(lt, batch):
        if capi._inited_device is None:
            capi.init(0)
        tables.append(capi.Table(lt.blob))
        return tables[-1].check(batch, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
    _lt, decided, flagged = _compare(_glob_store(80), _glob_inputs(80), run)
    assert decided >= 1 and flagged >= 16
    _lt, decided, flagged = _compare(_dr_store(70), _dr_inputs(70), run)
    assert decided == 64 and flagged == 6
    for t in tables:
        t.close()
