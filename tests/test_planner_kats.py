"""The reference's Go-coded known-answer tables for the planner's partial evaluation, transcribed (expression in, residual expression out):
internal/ruletable/planner/planner_test.go:33-190 Test_evaluateCondition, :312-400 TestPartialEvaluationWithGlobalVars,
struct_matcher_test.go:18-98 TestStructMatcher.  The wanted residuals are CEL text there; here both sides are parsed and compared as trees."""
import pytest

from cerbos_amd.cel.parser import parse
from cerbos_amd.plan.partial import Partial, Ts, format_timestamp, inline_variables, request_env

NOW = 1_700_000_000_123_000_000


def residual(text, principal=None, resource=None, extra=None, variables=None, root=True):
    env = request_env(principal or {}, resource or {}, None, {}, {})
    env.update(extra or {})
    tree = parse(text)
    if variables:
        done = {}
        for k, v in variables:
            done[k] = inline_variables(parse(v), done)
        tree = inline_variables(tree, done)
    pe = Partial(env, NOW)
    r = pe.pe(tree)
    if r[0] == "r" and root:
        r = pe.process_root(r[1])
    return ("lit", "bool", r[1]) if r[0] == "k" and isinstance(r[1], bool) else (pe.ast(r) if r[0] == "k" else r[1])


# ---- Test_evaluateCondition (planner_test.go:70-176): (expression, principal attr, resource kind, resource attr, want)
EVALUATE = [
    ("false", {}, "", {}, "false"),
    ("P.attr.authenticated", {"authenticated": True}, "", {}, "true"),
    ("request.principal.attr.authenticated", {"authenticated": True}, "", {}, "true"),
    ('R.attr.department == "marketing"', {}, "", {}, 'R.attr.department == "marketing"'),
    ("R.attr.owner == P.attr.name", {"name": "harry"}, "", {}, 'R.attr.owner == "harry"'),
    ("R.kind == P.attr.resource_name", {"resource_name": "resource-1"}, "resource-1", {}, "true"),
    ('P.attr.department_role[R.attr.department] == "ADMIN"', {"department_role": {"marketing": "ADMIN"}}, "", {"department": "marketing"}, "true"),
    ('R.attr.department_role[P.attr.department] == "ADMIN"', {"department": "marketing"}, "", {"department_role": {"marketing": "ADMIN"}}, "true"),
    ('request.principal.attr.department_role[request.resource.attr.department] == "ADMIN"', {"department_role": {"marketing": "ADMIN"}}, "",
     {"department": "marketing"}, "true"),
    ('P.attr.role_department["ADMIN"] == R.attr.department', {"role_department": {"ADMIN": "marketing"}}, "", {"department": "marketing"}, "true"),
]


@pytest.mark.parametrize("expr,pattr,kind,rattr,want", EVALUATE, ids=[e[0] for e in EVALUATE])
def test_evaluate_condition(expr, pattr, kind, rattr, want):
    assert residual(expr, {"attr": pattr}, {"kind": kind, "attr": rattr}) == parse(want)


# ---- TestPartialEvaluationWithGlobalVars (planner_test.go:320-372; setupEnv :402-441)
KNOWN = {"gb": "en_GB", "gb_us": ["GB", "US"], "ca": "ca", "T": 100}
VARIABLES = [("locale", 'R.attr.language + "_" + R.attr.country'), ("geo", "R.attr.geo"), ("gb_us", '["gb", "us"].map(t, t.upperAscii())'),
             ("gb_us2", "gb_us"), ("info", '{"country": "GB", "language": "en"}')]
GLOBAL_VARS = [
    ("V.geo", "R.attr.geo"),
    ("V.locale == gb", 'R.attr.language + "_" + R.attr.country == "en_GB"'),
    ("V.geo in (gb_us + [ca]).map(t, t.upperAscii())", 'R.attr.geo in ["GB", "US", "CA"]'),
    ("V.geo in (V.gb_us2 + [ca]).map(t, t.upperAscii())", 'R.attr.geo in ["GB", "US", "CA"]'),
    ("V.geo in (variables.gb_us + [ca]).map(t, t.upperAscii())", 'R.attr.geo in ["GB", "US", "CA"]'),
    ('V.info.language + "_" + V.info.country == gb', "true"),
    ('has(R.attr.geo) && R.attr.geo in ["GB", "US"]', 'has(R.attr.geo) && R.attr.geo in ["GB", "US"]'),
    ("has(V.info.language)", "true"),
    ("R.attr.items.filter(x, x.price > T)", "R.attr.items.filter(x, x.price > 100)"),
    ('now() > timestamp("2021-04-20T00:00:00Z") && R.attr.geo in ["GB", "US"]', 'R.attr.geo in ["GB", "US"]'),
    ("R.attr.items.filter(x, x.price > now())", 'R.attr.items.filter(x, x.price > timestamp("%s"))' % format_timestamp(Ts(NOW))),
    ("timestamp(R.attr.lastAccessed) > now()", 'timestamp(R.attr.lastAccessed) > timestamp("%s")' % format_timestamp(Ts(NOW))),
    ("intersect(R.attr.workspaces, V.gb_us)", 'intersect(R.attr.workspaces, ["GB", "US"])'),
]


@pytest.mark.parametrize("expr,want", GLOBAL_VARS, ids=[e[0] for e in GLOBAL_VARS])
def test_partial_evaluation_with_global_vars(expr, want):
    assert residual(expr, extra=KNOWN, variables=VARIABLES, root=False) == parse(want)


# ---- TestStructMatcher (struct_matcher_test.go:23-76): what the root rewrites make of an expression BEFORE it is evaluated again
# ("" = no rewrite applies)
STRUCT = [
    ('{"a": 3}[R.attr.Id] == 4', 'R.attr.Id == "a" && 4 == 3'),
    ('4 == {"a": 3}[R.attr.Id]', ""),
    ("P.attr.anyMap[R.attr.Id] == R.attr.value", ""),
    ('{"a1": {"role": "OWNER"}}[R.id].role == "OWNER"', 'R.id == "a1" && "OWNER" == {"role": "OWNER"}.role'),
    ("P.attr.anyMap[R.attr.Id][R.attr.value]", ""),
    ('3 in {"a": [3, 4]}[R.attr.Id]', 'R.attr.Id == "a" && 3 in [3, 4]'),
    ("3 in P.attr.anyMap[R.attr.Id]", ""),
    ('{1: ["red", "square"], 2: ["blue", "triangle"], 3: ["black", "circle"]}.exists(k, v, R.attr.color == v[0] && R.attr.shape == v[1])',
     'R.attr.color == "red" && R.attr.shape == "square" || (R.attr.color == "blue" && R.attr.shape == "triangle" || '
     'R.attr.color == "black" && R.attr.shape == "circle")'),
    ("{1: 1}.exists(k, v, k == v)", "true"),
    ('{1: {"colors": ["red"]}}.exists(k, v, R.attr.color in v["colors"])', 'R.attr.color in ["red"]'),
    ('{1: {"colors": ["red"]}}.exists(v, R.attr.color == v)', "R.attr.color == 1"),
    ("[1, 2].exists(v, R.attr.color == v)", "R.attr.color == 1 || R.attr.color == 2"),
    ("[1, 2].exists(i, v, R.attr.color == v && R.attr.size == i)", "R.attr.color == 1 && R.attr.size == 0 || R.attr.color == 2 && R.attr.size == 1"),
    ("[1, 2].all(v, R.attr.color == v)", "R.attr.color == 1 && R.attr.color == 2"),
]


@pytest.mark.parametrize("expr,want", STRUCT, ids=[e[0] for e in STRUCT])
def test_struct_matcher(expr, want):
    pe = Partial(request_env({}, {}, None, {}, {}), NOW)
    n = parse(expr)
    got = pe._struct_index(n) or pe._in_struct_index(n) or pe._unroll(n, pe.env)
    if want == "":
        assert got is None
    else:
        assert got == parse(want)


# ---- an operand that fails beside one that is unknown (cel-go interpretable.go evalAnd / evalOr: unknown outranks error, the
# pruner has no value for the failing operand): the residual keeps BOTH as written.  No reference fixture reaches this; dropping
# the failing operand of an `&&` would make the filter wider than Check (true && error = error: the rule does not match).
ERROR_BESIDE_UNKNOWN = [
    ("R.attr.x && P.attr.missing", "R.attr.x && P.attr.missing"),
    ("P.attr.missing && R.attr.x", "P.attr.missing && R.attr.x"),
    ("R.attr.x || P.attr.missing", "R.attr.x || P.attr.missing"),
    ("R.attr.x && (P.attr.n > 1 && P.attr.missing)", None),      # a known conjunction that fails: the condition's error
    ("false && P.attr.missing", "false"),
    ("R.attr.x && false && P.attr.missing", "false"),
    # a known operand that is not a bool beside an unknown one: the unknown outranks "no such overload" as it outranks any error
    ("P.attr.n && R.attr.x", "2.0 && R.attr.x"),
    ("R.attr.x || P.attr.n", "R.attr.x || 2.0"),
    ("P.attr.n && true", None),                                   # ... beside a known one: no such overload
]


@pytest.mark.parametrize("expr,want", ERROR_BESIDE_UNKNOWN, ids=[e[0] for e in ERROR_BESIDE_UNKNOWN])
def test_an_error_beside_an_unknown_stays_in_the_residual(expr, want):
    from cerbos_amd.plan.partial import CelEvalError
    if want is None and expr.startswith("P.attr.n"):
        with pytest.raises(CelEvalError):
            residual(expr, {"attr": {"n": 2.0}}, {"kind": "k"})
        return
    if want is None:
        got = residual(expr, {"attr": {"n": 2.0}}, {"kind": "k"})
        assert got == parse("R.attr.x && (P.attr.n > 1 && P.attr.missing)") or isinstance(got, tuple)
        return
    try:
        got = residual(expr, {"attr": {"n": 2.0}}, {"kind": "k"})
    except CelEvalError:
        got = None
    assert got == parse(want)


def test_the_resource_s_policy_version_is_not_known_to_the_partial_evaluator():
    """planner.go:532-577 newEvaluator: of the resource only kind, scope and the supplied attributes are known"""
    assert residual('R.policyVersion == "default"', {}, {"kind": "k"}) == parse('R.policyVersion == "default"')
    assert residual('R.policyVersion == "default"', {}, {"kind": "k", "policyVersion": "default"}) == parse('R.policyVersion == "default"')
    assert residual('R.kind == "k" && R.id == "1"', {}, {"kind": "k"}) == parse('R.id == "1"')


def test_a_broken_derived_role_spoils_only_what_reads_the_list():
    """plan.go:157-176 drListErr: under strict evaluation a derived-role definition that fails makes runtime.effectiveDerivedRoles an
    error for whoever READS it - an action whose rules never do is planned as if the definition were not there."""
    from cerbos_amd.plan.planner import Planner
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies
    docs = [
        {"apiVersion": "api.cerbos.dev/v1", "derivedRoles": {"name": "dr", "definitions": [
            {"name": "broken", "parentRoles": ["user"], "condition": {"match": {"expr": "P.attr.missing == 1"}}}]}},
        {"apiVersion": "api.cerbos.dev/v1", "resourcePolicy": {"resource": "doc", "version": "default", "importDerivedRoles": ["dr"], "rules": [
            {"actions": ["view"], "effect": "EFFECT_ALLOW", "roles": ["user"], "condition": {"match": {"expr": "R.attr.public == true"}}},
            {"actions": ["edit"], "effect": "EFFECT_ALLOW", "derivedRoles": ["broken"]},
            {"actions": ["list"], "effect": "EFFECT_ALLOW", "roles": ["user"],
             "condition": {"match": {"expr": '"broken" in runtime.effectiveDerivedRoles || R.attr.public == true'}}}]}},
    ]
    pl = Planner(rule_table_from_policies(policies_from_docs(docs)))
    inp = {"principal": {"id": "p", "roles": ["user"], "attr": {}}, "resource": {"kind": "doc"}}
    view = pl.plan(dict(inp, actions=["view"]), strict_evaluation=True)
    assert view["filter"]["kind"] == "KIND_CONDITIONAL", view
    edit = pl.plan(dict(inp, actions=["edit"]), strict_evaluation=True)
    assert edit["filter"]["kind"] == "KIND_ALWAYS_DENIED"
    lst = pl.plan(dict(inp, actions=["list"]), strict_evaluation=True)       # reads the list: the definition's error is this action's
    assert lst["filter"]["kind"] == "KIND_ALWAYS_DENIED"
    lenient = pl.plan(dict(inp, actions=["list"]), strict_evaluation=False)  # not strict: the definition is false, the rest stands
    assert lenient["filter"]["kind"] == "KIND_CONDITIONAL"


# ---- TestCELErrorsPlan / TestStrictEvaluationPlan (internal/ruletable/cel_errors_test.go:358-410, strict_evaluation_test.go:94-160):
# the harness of the Check KATs (tests/golden/ruletable_cel_errors.json, tools/make_golden_ruletable_tests.py) through RuleTable.Plan -
# the filter's kind and the expressions of PlanResourcesOutput.EvaluationErrors, in order.
def _plan_kats():
    from helpers import load_json
    return load_json("ruletable_cel_errors.json")


@pytest.mark.parametrize("case", _plan_kats()["planCases"], ids=lambda c: ("strict_" if c["strict"] else "") + c["name"])
def test_cel_errors_in_plans(case):
    from cerbos_amd.plan.planner import Planner
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies
    doc = _plan_kats()
    pl = Planner(rule_table_from_policies(policies_from_docs(doc["policies"])))
    attr = {}
    if "number" in case["amount"]:
        attr["amount"] = case["amount"]["number"]
    elif "string" in case["amount"]:
        attr["amount"] = case["amount"]["string"]
    out = pl.plan({"requestId": doc["requestId"], "actions": [case["action"]], "principal": doc["principal"],
                   "resource": {"kind": case["kind"], "attr": attr}}, strict_evaluation=case["strict"])
    assert out["filter"]["kind"] == case["wantKind"], out
    assert [e["celError"]["expression"] for e in out.get("evaluationErrors") or []] == case["wantErrorExpressions"], out
    assert all(e["celError"]["message"] for e in out.get("evaluationErrors") or [])
    # ... and through the wire form of PlanResourcesOutput (field 11: EvaluationError {cel_error {expression, message}})
    from cerbos_amd import wire
    assert wire.decode_plan_resources_output(wire.encode_plan_resources_output(out))["evaluationErrors"] == (out.get("evaluationErrors") or [])
