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
