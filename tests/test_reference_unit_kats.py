"""Go-coded unit tests of the reference that sit on the hot path, transcribed by hand into tests/golden/reference_unit_kats.json
(they are code, not tables) and answered by the oracle, by the product's compile-time checks and - where an expectation can be
put as a rule's condition - by the DEVICE path (lowering + the kernel source on the host simulator):

* internal/conditions/types/registry_test.go - TestJSONFields (request.aux_data == request.auxData, `undefined field` for a
  field the message does not have), TestRuntime (runtime.effective_derived_roles under both names), TestVariables (V.x present /
  absent): SURVEY §8 rows a12 / a13;
* internal/conditions/cel_test.go - TestExpandAbbrev, TestResourceAttributeNames: R / P / V / G are the long names;
* internal/ruletable/index/index_test.go TestParentRoleIndex - AddParentRoles over the ancestors compiled per scope: row a8;
* internal/ruletable/internal/utils_test.go TestSetIntersects - a derived role's parent roles against the principal's roles
  (`*` is a wildcard, `foo*` is not a pattern): rows a8 / a12 (check.go:244);
* internal/evaluator/cel_errors_test.go TestCELErrorsAdd - identical (expression, message) pairs are kept once: row a15."""
import pytest

from cerbos_amd.cel.check import compile_issues
from cerbos_amd.engine import Conf
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.lower.celc import LoweringError
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import _compile_parent_role_ancestors, rule_table_from_policies
from helpers import load_json
from oracle import celeval
from oracle.check import EvalParams, Index, RuleTableOracle, _EvalContext
from test_hostsim_golden import HostSimEvaluator

KATS = load_json("reference_unit_kats.json")
API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000
INPUT = {"requestId": "kat", "principal": {"id": "p", "roles": ["user"], "attr": {"department": "marketing"}},
         "resource": {"kind": "kat", "id": "r", "attr": {"department": "marketing"}}, "actions": ["a"],
         "auxData": {"jwt": {"fooBar": "baz"}}}


def _env(bindings):
    ev = _EvalContext(EvalParams(now_ns=NOW, globals_={"environment": "test"}), INPUT)
    if bindings == "runtime":   # registry_test.go:72-79: the runtime's roles, in the order the test gives them
        ev._runtime = celeval.Message({"effective_derived_roles": ["foo", "bar"]}, {"effectiveDerivedRoles": "effective_derived_roles"})
    return ev._env({}, {"foo": "bar"} if bindings == "V" else {"is_admin": True})


def _native(v):
    return list(v) if isinstance(v, (list, tuple)) else v


@pytest.mark.parametrize("case", KATS["cel"], ids=lambda c: "%s/%s" % (c["test"], c["name"]))
def test_cel_case_by_the_oracle_and_the_compile_checks(case):
    ast, msgs = compile_issues(case["expr"])
    if "want_compile_err" in case:
        assert any(case["want_compile_err"] in m for m in msgs), msgs
        return
    assert ast is not None and msgs == [], msgs
    if "want_eval_err" in case:
        with pytest.raises(celeval.CelError) as e:
            celeval.evaluate(case["expr"], _env(case["bindings"]))
        assert case["want_eval_err"] in str(e.value)
        return
    assert _native(celeval.evaluate(case["expr"], _env(case["bindings"]))) == case["want_result"]


def _device_decides(docs, inp, action="a"):
    """The request through lowering + kernel source on the simulator: ALLOW? (None = flagged outside the device subset)"""
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(docs)))
    outs, bad = HostSimEvaluator(lt, Conf()).check([inp], now_ns=NOW, allow_unsupported=True)
    return None if bad else outs[0]["actions"][action]["effect"] == "EFFECT_ALLOW"


def _policy(expr, variables=None):
    rp = {"resource": "kat", "version": "default",
          "rules": [{"actions": ["a"], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": expr}}}]}
    if variables:
        rp["variables"] = {"local": variables}
    return [{"apiVersion": API, "resourcePolicy": rp}]


def test_json_and_proto_names_on_the_device_path():
    """TestJSONFields' expectations as conditions: the snake-case and the camel-case name read the same field."""
    decided = 0
    for case in KATS["cel"]:
        if case["test"] != "TestJSONFields" or "want_result" not in case:
            continue
        expr = case["expr"] if case["want_result"] is True else "%s == %r" % (case["expr"], case["want_result"])
        try:
            got = _device_decides(_policy(expr.replace("'", '"')), INPUT)
        except LoweringError:
            continue
        if got is not None:
            assert got is True, expr
            decided += 1
    assert decided >= 2   # the two reads of the claim


def test_variables_on_the_device_path():
    """TestVariables: V.foo present - its value, has(V.foo); an undefined V.bar never reaches evaluation: the policy compiler
    rejects the policy (compile.go: undefined variable), as the reference's does."""
    assert _device_decides(_policy('V.foo == "bar" && has(V.foo)', {"foo": '"bar"'}), INPUT) in (True, None)
    assert _device_decides(_policy('V.foo == "bar"', {"foo": '"bar"'}), INPUT) is True
    with pytest.raises(Exception) as e:
        _device_decides(_policy('V.bar == "bar"', {"foo": '"bar"'}), INPUT)
    assert "bar" in str(e.value)


def test_expand_abbrev_pairs_evaluate_alike():
    """TestExpandAbbrev / TestResourceAttributeNames: the abbreviation and the long name are one value - in the oracle, and as
    a condition `short == long` on the device path."""
    ran = [KATS["resource_attribute_names"]["want"]]
    assert ran[0][0] == "R.attr.%s" % KATS["resource_attribute_names"]["name"]
    for short, long_ in KATS["expand_abbrev"]:
        env = _env(None)
        a, b = celeval.evaluate(short, env), celeval.evaluate(long_, env)
        if isinstance(a, (celeval.Message, celeval.Variables)):
            assert a is b, (short, long_)     # the same binding (cel.go:42-53 declares both names over one value)
        else:
            assert a == b, (short, long_)
    on_device = 0
    for short, long_ in KATS["expand_abbrev"]:
        if "." not in short or short == long_:
            continue
        variables = {"is_admin": "true"} if short.startswith("V.") else None
        ev_conf = Conf(globals_={"environment": "test"}) if short.startswith("G.") else Conf()
        lt = lower_rule_table(rule_table_from_policies(policies_from_docs(_policy("%s == %s" % (short, long_), variables))), ev_conf.globals or None)
        outs, bad = HostSimEvaluator(lt, ev_conf).check([INPUT], now_ns=NOW, allow_unsupported=True)
        if not bad:
            assert outs[0]["actions"]["a"]["effect"] == "EFFECT_ALLOW", (short, long_)
            on_device += 1
    assert on_device >= 3


def test_parent_role_index():
    kat = KATS["parent_role_index"]
    for case in kat["cases"]:
        spr = case.get("replace_with", kat["scope_parent_roles"])
        idx = Index({"rules": [], "parent_roles": _compile_parent_role_ancestors(spr)})
        assert sorted(idx.add_parent_roles(case["scopes"], case["roles"])) == sorted(case["want"]), case["name"]


def _derived_role_store(parent_roles):
    return [{"apiVersion": API, "derivedRoles": {"name": "drs", "definitions": [{"name": "dr", "parentRoles": parent_roles}]}},
            {"apiVersion": API, "resourcePolicy": {"resource": "kat", "version": "default", "importDerivedRoles": ["drs"],
                                                   "rules": [{"actions": ["a"], "derivedRoles": ["dr"], "effect": "EFFECT_ALLOW"}]}}]


@pytest.mark.parametrize("case", [c for c in KATS["set_intersects"] if c["s1"]], ids=lambda c: c["name"])
def test_set_intersects_as_a_derived_role_activates(case):
    """SetIntersects(parent roles of the definition, roles of the principal) decides whether a derived role is considered
    (check.go:244): the oracle and the device path on a store whose only ALLOW needs that derived role."""
    docs = _derived_role_store(case["s1"])
    inp = dict(INPUT, principal=dict(INPUT["principal"], roles=case["s2"]))
    rt = rule_table_from_policies(policies_from_docs(docs))
    want = RuleTableOracle(rt).check(inp, EvalParams(now_ns=NOW))
    assert (want["actions"]["a"]["effect"] == "EFFECT_ALLOW") == case["want"]
    assert ("dr" in (want.get("effectiveDerivedRoles") or [])) == case["want"]
    lt = lower_rule_table(rt)
    outs, bad = HostSimEvaluator(lt, Conf()).check([inp], now_ns=NOW, allow_unsupported=True)
    assert not bad
    assert (outs[0]["actions"]["a"]["effect"] == "EFFECT_ALLOW") == case["want"]
    assert ("dr" in outs[0]["effectiveDerivedRoles"]) == case["want"]


def test_cel_errors_are_kept_once():
    kat = KATS["cel_errors_dedup"]
    ev = _EvalContext(EvalParams(now_ns=NOW), INPUT)
    for expr, msg in kat["adds"]:
        ev.add_error(expr, celeval.CelError(msg))
    errs = ev.all_errors()
    assert len(errs) == kat["want_len"]
    assert [(e["celError"]["expression"], e["celError"]["message"]) for e in errs] == sorted({tuple(a) for a in kat["adds"]})
