"""PlanResources against the reference's query-planner suites (internal/test/testdata/query_planner: 27 suites, 116 plans;
engine_test.go:420-495 TestQueryPlan): the filter of every (principal, action(s), resource kind) compared as the reference compares
it - operands of every expression in any order (protocmp.SortRepeatedFields on "operands", engine_test.go:486-488)."""
import json

import pytest

from cerbos_amd.plan import Planner
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json

DATA = load_json("planner_cases.json")
NOW = 1705353507395000000            # 2024-01-16T10:18:27.395+13:00 (engine_test.go:448)
AUX = {"jwt": {"customInt": 42}}     # engine_test.go:424-425
CASES = [(s, k) for s in DATA["suites"] for k in range(len(s["tests"]))]


@pytest.fixture(scope="module")
def planner():
    return Planner(rule_table_from_policies(policies_from_docs(DATA["policies"])))


def canon(op):
    """Operands in any order, numbers as doubles."""
    if op is None:
        return None
    if "expression" in op:
        e = op["expression"]
        kids = sorted((canon(o) for o in e.get("operands") or []), key=lambda x: json.dumps(x, sort_keys=True))
        return {"expression": {"operator": e["operator"], "operands": kids}}
    if "value" in op:
        return {"value": _num(op["value"])}
    return {"variable": op["variable"]}


def _num(v):
    if isinstance(v, bool) or v is None or isinstance(v, str):
        return v
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, list):
        return [_num(x) for x in v]
    return {k: _num(x) for k, x in v.items()}


def run(planner, suite, test, lenient):
    inp = {"requestId": "requestId", "principal": suite["principal"], "resource": test["resource"], "actions": test["actions"], "auxData": AUX}
    return planner.plan(inp, globals_={"environment": "test"}, lenient_scope_search=lenient, now_ns=NOW)   # engine_test.go mkEngine: globals


@pytest.mark.parametrize("suite,k", CASES, ids=["%s-%d" % (s["name"], k) for s, k in CASES])
def test_plan(planner, suite, k):
    test = suite["tests"][k]
    for lenient in ([False, True] if suite["lenient"] is None else [suite["lenient"]]):
        if test["wantErr"]:
            with pytest.raises(Exception):
                run(planner, suite, test, lenient)
            continue
        have = run(planner, suite, test, lenient)["filter"]
        want = test["want"]
        assert have["kind"] == want.get("kind"), (test["actions"], have)
        assert canon(have.get("condition")) == canon(want.get("condition")), (test["actions"], json.dumps(have.get("condition")))


@pytest.mark.parametrize("case", DATA["filters"], ids=[c["name"] for c in DATA["filters"]])
def test_normalise_filter(case):
    """TestNormaliseFilter (planner_test.go:443-456): exact, operands in order, and the debug string"""
    from cerbos_amd.plan import filter as flt
    have = flt.normalise_filter(case["input"])
    want = case["wantFilter"]
    assert have.get("kind") == want.get("kind")
    assert _exact(have.get("condition")) == _exact(want.get("condition")), json.dumps(have)
    assert flt.filter_to_string(have) == case["wantString"]


def _exact(op):
    if op is None:
        return None
    if "expression" in op:
        return {"expression": {"operator": op["expression"]["operator"], "operands": [_exact(o) for o in op["expression"].get("operands") or []]}}
    if "value" in op:
        return {"value": _num(op["value"])}
    return {"variable": op["variable"]}


@pytest.mark.parametrize("text", sorted(DATA["buildExpr"]), ids=sorted(DATA["buildExpr"]))
def test_build_expr(text):
    """Test_buildExpr (ast_test.go:49-70): CEL text -> filter operand, exactly"""
    from cerbos_amd.cel.parser import parse
    from cerbos_amd.plan import filter as flt
    assert _exact(flt.build(parse(text))) == _exact(DATA["buildExpr"][text]), json.dumps(flt.build(parse(text)))


# ---- the service-level cases (internal/test/testdata/server/plan_resources; svc/cerbos_svc.go:53-118) over the engine store
SERVER = [c for c in DATA["serverPlans"] if "validationErrors" not in c["wantResponse"]]     # (schema validation is outside the path)


@pytest.fixture(scope="module")
def store_planner():
    from helpers import store_rule_table
    return Planner(store_rule_table())


@pytest.mark.parametrize("case", SERVER, ids=[c["name"] for c in SERVER])
def test_service_level_plan(store_planner, case):
    from cerbos_amd.plan import plan_resources_response
    have = plan_resources_response(store_planner, case["input"], globals_={"environment": "test"}, now_ns=NOW)
    want = case["wantResponse"]
    assert have["filter"]["kind"] == want["filter"]["kind"]
    assert canon(have["filter"].get("condition")) == canon(want["filter"].get("condition")), json.dumps(have["filter"])
    for k in ("requestId", "action", "actions", "resourceKind", "policyVersion"):
        assert (have.get(k) or None) == (want.get(k) or None), k
    hm, wm = have.get("meta") or {}, want.get("meta") or {}
    assert (hm.get("matchedScope") or "") == (wm.get("matchedScope") or "") and (hm.get("matchedScopes") or {}) == (wm.get("matchedScopes") or {})
    if want["filter"]["kind"] != "KIND_CONDITIONAL":
        assert hm.get("filterDebug") == wm.get("filterDebug")
    else:   # the debug string prints operands in the order the walk met them: compare what it says, not how
        assert sorted(hm["filterDebug"].replace("(", " ( ").replace(")", " ) ").split()) == sorted(wm["filterDebug"].replace("(", " ( ").replace(")", " ) ").split())
