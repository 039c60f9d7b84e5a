"""The oracle's index query (oracle/check.py Index.query - SURVEY §8 rows a6 / a7: `Index.Query` with the role-policy DENY
synthesis of index.go:352-598) against the reference's own unit tests of it: internal/ruletable/index/index_test.go
TestQueryAllowActionsSyntheticDeny and the one-resource / one-action subtests of TestQueryMultiSynthesis, transcribed by hand
into tests/golden/index_query_cases.json (they are Go code, not tables: no script can mine them)."""
import pytest

from cerbos_amd import namer
from cerbos_amd.policy.compile import SP_REQUIRE_PARENTAL_CONSENT
from cerbos_amd.ruletable.build import KIND_RESOURCE, _blank_row
from helpers import load_json
from oracle.check import Index

CASES = load_json("index_query_cases.json")["cases"]
_SP = {"SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS": SP_REQUIRE_PARENTAL_CONSENT}


def _cond(c):
    if c is None:
        return None
    if "expr" in c:
        return ("expr", c["expr"])
    (op, items), = c.items()
    return (op, tuple(_cond(x) for x in items))


def _table(rows):
    out = []
    for n, r in enumerate(rows):
        assert r["role_policy"]
        out.append(_blank_row(id=n, origin_fqn=namer.role_policy_fqn(r["role"], "default", ""), role=r["role"], resource=r["resource"],
                              allow_actions=list(r["allow_actions"]), condition=_cond(r["condition"]), name=r["name"], effect=r["effect"],
                              version="default", scope="", policy_kind=KIND_RESOURCE, from_role_policy=True))
    return {"rules": out, "parent_roles": {}}


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_query_as_the_reference_tests_it(case):
    q = case["query"]
    got = Index(_table(case["rows"])).query("default", q["resource"], "", q["action"], q["roles"], KIND_RESOURCE, "")
    denies = [b for b in got if b["effect"] == "DENY"]
    assert len(got) == len(denies) == len(case["want"]), got
    for b, w in zip(denies, case["want"]):
        for key, val in w.items():
            if key == "condition":
                assert b["condition"] == _cond(val)
            elif key == "scope_permissions":
                assert b["scope_permissions"] == _SP[val]
            else:
                assert b[key] == val, (key, b)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_the_planner_s_query_on_the_same_cases(case):
    """cerbos_amd/plan/index.py PlanIndex.query (the product's host-side Index.Query, used by PlanResources) on the same transcribed cases."""
    from cerbos_amd.plan.index import PlanIndex
    q = case["query"]
    got = PlanIndex(_table(case["rows"])).query("default", q["resource"], "", q["action"], q["roles"], KIND_RESOURCE, "")
    denies = [b for b in got if b["effect"] == "DENY"]
    assert len(got) == len(denies) == len(case["want"]), got
    for b, w in zip(denies, case["want"]):
        for key, val in w.items():
            if key == "condition":
                assert b["condition"] == _cond(val)
            elif key == "scope_permissions":
                continue      # (the planner reads a scope's permissions from the table, not from the binding)
            else:
                assert b[key] == val, (key, b)


@pytest.mark.parametrize("seed", range(6))
def test_the_planner_s_query_against_the_oracle_s_on_generated_stores(seed):
    """Two restatements of Index.Query written apart (oracle/check.py for the checker, cerbos_amd/plan/index.py for PlanResources):
    the same bindings in the same order for every (scope, action, role set, policy kind, principal) of generated stores."""
    import numpy as np

    from cerbos_amd.plan.index import PlanIndex
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import KIND_PRINCIPAL, rule_table_from_policies
    from test_fuzz_parity import ACTIONS, KINDS, ROLES, SCOPES, _policies
    rng = np.random.default_rng(900 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    a, b = Index(rt), PlanIndex(rt)
    n = 0

    def key(x):
        return (None if x.get("from_role_policy") else x.get("id"), x["effect"], x["origin_fqn"], x["role"], x["action"], x["condition"], x.get("derived_role_condition"), bool(x.get("from_role_policy")))
    for version in ("default", "v2"):
        for kind in KINDS:
            res = namer.sanitize(kind)
            for scope in SCOPES:
                for action in ACTIONS + ["act0"]:
                    for roles in ([r] for r in ROLES):
                        rs = b.add_parent_roles([scope], roles)
                        assert rs == a.add_parent_roles([scope], roles)
                        for pk, pid in ((KIND_RESOURCE, ""), (KIND_PRINCIPAL, "p0"), (KIND_PRINCIPAL, "p3")):
                            x, y = a.query(version, res, scope, action, rs, pk, pid), b.query(version, res, scope, action, rs, pk, pid)
                            assert [key(i) for i in x] == [key(i) for i in y], (version, kind, scope, action, roles, pk, pid)
                            n += bool(x)
    assert n > 20
