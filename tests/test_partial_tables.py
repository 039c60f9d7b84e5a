"""Rules the device cannot evaluate faithfully do not take the table down with them: the lowering turns their
conditions into UNSUPPORTED programs, the requests that reach them are flagged for the caller's own engine, every
other request is decided as the oracle decides it.

* a principal-policy condition reading runtime.effectiveDerivedRoles (check.go:281: the value depends on the
  previously evaluated action);
* role-policy rules for overlapping resource globs that share an evaluation key but not a condition
  (ruletable.go:445-455: the per-request condition cache then serves one rule the other's outcome)."""
from cerbos_amd.engine import Conf
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000


def _decide(docs, inputs):
    from test_hostsim_golden import HostSimEvaluator
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    outs, bad = HostSimEvaluator(lt, Conf()).check(inputs, now_ns=NOW, allow_unsupported=True)
    orc = RuleTableOracle(rt)
    for i, (inp, have) in enumerate(zip(inputs, outs)):
        if i not in bad:
            assert norm_actions(have) == norm_actions(orc.check(inp, EvalParams(now_ns=NOW))), inp
    return lt, bad


def test_principal_policy_reading_runtime_flags_only_its_own_requests():
    docs = [
        {"apiVersion": API, "derivedRoles": {"name": "dr", "definitions": [{"name": "owner", "parentRoles": ["user"],
                                                                             "condition": {"match": {"expr": "R.attr.owner == P.id"}}}]}},
        {"apiVersion": API, "resourcePolicy": {"resource": "doc", "version": "default", "importDerivedRoles": ["dr"], "rules": [
            {"actions": ["view"], "roles": ["user"], "effect": "EFFECT_ALLOW"},
            {"actions": ["edit"], "derivedRoles": ["owner"], "effect": "EFFECT_ALLOW"}]}},
        {"apiVersion": API, "principalPolicy": {"principal": "alice", "version": "default", "rules": [
            {"resource": "doc", "actions": [{"action": "edit", "effect": "EFFECT_DENY",
                                             "condition": {"match": {"expr": '"owner" in runtime.effectiveDerivedRoles'}}}]}]}},
    ]
    mk = lambda pid, actions: {"requestId": pid, "actions": actions, "principal": {"id": pid, "roles": ["user"]},   # noqa: E731
                               "resource": {"kind": "doc", "id": "d1", "attr": {"owner": pid}}}
    inputs = [mk("bob", ["view", "edit"]), mk("alice", ["view"]), mk("alice", ["edit"]), mk("carol", ["edit"])]
    lt, bad = _decide(docs, inputs)
    assert lt.unsupported and "runtime.effectiveDerivedRoles" in lt.unsupported[0][1]
    assert bad == [2]   # only alice's `edit` reaches the rule


def test_history_dependent_role_policy_rules_flag_only_their_requests():
    docs = [
        {"apiVersion": API, "resourcePolicy": {"resource": "leave_request", "version": "default", "rules": [
            {"actions": ["view", "approve"], "roles": ["employee", "manager"], "effect": "EFFECT_ALLOW"}]}},
        {"apiVersion": API, "resourcePolicy": {"resource": "report", "version": "default", "rules": [
            {"actions": ["view"], "roles": ["employee", "manager"], "effect": "EFFECT_ALLOW"}]}},
        {"apiVersion": API, "rolePolicy": {"role": "manager", "rules": [
            {"resource": "leave_*", "allowActions": ["view"], "condition": {"match": {"expr": "R.attr.team == P.attr.team"}}},
            {"resource": "*", "allowActions": ["view"], "condition": {"match": {"expr": "R.attr.public == true"}}}]}},
    ]
    mk = lambda role, kind: {"requestId": role + kind, "actions": ["view"], "principal": {"id": "p", "roles": [role], "attr": {"team": "a"}},   # noqa: E731
                             "resource": {"kind": kind, "id": "r", "attr": {"team": "a", "public": False}}}
    inputs = [mk("employee", "leave_request"), mk("employee", "report"), mk("manager", "leave_request"), mk("manager", "report")]
    lt, bad = _decide(docs, inputs)
    assert lt.unsupported and "evaluation key" in lt.unsupported[0][1]
    assert 0 not in bad and 1 not in bad and bad   # the employee's requests never meet the role policy; the manager's are flagged
