"""The policy compiler (cerbos_amd/policy/compile.py - SURVEY §8 row a11's compile stage) against the reference's own compiler
cases: internal/test/testdata/compile/*.yaml run by TestCompile (internal/compile/compile_test.go:40-88), mined into
tests/golden/compile_cases.json by tools/make_golden_compile.py.

* a case with `wantErrors`: the same SET of errors - file, error kind, description, JSON path, line and column.  For syntax
  errors the text inside `[...]` is ANTLR's and is not compared; the two schema cases need the schema loader (out of scope);
* a case with a golden: the RunnablePolicySet the reference's compiler produced (protojson), field by field, without cel-go's
  type-checked expression trees."""
import pytest

from cerbos_amd import namer
from cerbos_amd.policy import compile as pc
from cerbos_amd.policy.loader import policy_fqn, policy_kind
from cerbos_amd.policy.source import load_yaml_with_source
from helpers import load_json

CASES = load_json("compile_cases.json")["cases"]
NEEDS_SCHEMA_LOADER = ("invalid_schemas", "missing_schemas")
_SP = {pc.SP_UNSPECIFIED: "SCOPE_PERMISSIONS_OVERRIDE_PARENT", pc.SP_OVERRIDE_PARENT: "SCOPE_PERMISSIONS_OVERRIDE_PARENT",
       pc.SP_REQUIRE_PARENTAL_CONSENT: "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS"}


def _unit(case):
    policies, sources, files = {}, {}, {}
    for file, text in case["files"].items():
        doc, src = load_yaml_with_source(text, file)
        fqn = policy_fqn(doc)
        policies[fqn], sources[fqn], files[fqn] = doc, src, file
    main = policy_fqn(load_yaml_with_source(case["files"][case["mainDef"]], case["mainDef"])[0])
    return policies, sources, files, main


@pytest.mark.parametrize("case", [c for c in CASES if c["wantErrors"]], ids=lambda c: c["name"])
def test_rejected_with_the_reference_errors(case):
    if case["name"] in NEEDS_SCHEMA_LOADER:
        pytest.skip("schema references are not loaded (schema validation is out of scope)")
    policies, sources, _, main = _unit(case)
    with pytest.raises(pc.CompileError) as got:
        pc.compile_unit(policies, main, sources)
    have = got.value.errors
    want = case["wantErrors"]
    assert len(have) == len(want), "\n".join(map(str, have))

    def norm(file, error, desc, path, line, col):
        if "Syntax error" in desc:   # this parser's message differs from ANTLR's; the rest of the report must not
            desc = desc[:desc.index("[")]
        return (file, error, desc, path or "", line or 0, col or 0)
    want_set = sorted(norm(w["file"], w["error"], w.get("description", ""), (w.get("position") or {}).get("path"),
                           (w.get("position") or {}).get("line"), (w.get("position") or {}).get("column")) for w in want)
    assert sorted(norm(*e) for e in have) == want_set


def test_rejected_without_the_yaml_text_too():
    """Without positions (policies as dicts) the same errors come out, descriptions in the reference's no-position form."""
    for case in CASES:
        if not case["wantErrors"] or case["name"] in NEEDS_SCHEMA_LOADER:
            continue
        policies, _, _, main = _unit(case)
        with pytest.raises(pc.CompileError) as got:
            pc.compile_unit(policies, main)
        # an error is identified by file, position and description (errors.go:105-121): without positions equal texts fold
        # into one, and the texts that quote positions (redefinitions, cycles, ambiguous imports) take their short form
        assert {e.error for e in got.value.errors} == {w["error"] for w in case["wantErrors"]}, case["name"]
        plain = lambda d: "Syntax" not in d and ".yaml" not in d and "form a cycle" not in d   # noqa: E731
        assert {e.description for e in got.value.errors if plain(e.description)} == \
            {w["description"] for w in case["wantErrors"] if plain(w["description"])}, case["name"]
        assert all(e.line is None for e in got.value.errors)
        if case["name"] == "bad_variables":
            assert {"Variables 'b' and 'c' form a cycle", "Variables 'd', 'e', 'f', and 'g' form a cycle"} <= {e.description for e in got.value.errors}


# ---- the runnable policy set as protojson prints it (api/public/cerbos/runtime/v1/runtime.proto) ------------------------------
def _expr(text):
    return {"original": text}


def _cond(c):
    if c is None:
        return None
    if c[0] == "expr":
        return {"expr": _expr(c[1])}
    return {c[0]: {"expr": [_cond(x) for x in c[1]]}}


def _put(d, key, value):
    if value not in (None, "", {}, []):
        d[key] = value


def _output(when):
    if when is None:
        return None
    w = {}
    _put(w, "ruleActivated", _expr(when["rule_activated"]) if "rule_activated" in when else None)
    _put(w, "conditionNotMet", _expr(when["condition_not_met"]) if "condition_not_met" in when else None)
    return {"when": w}


def _vars(d, compiled, with_map=True):
    _put(d, "orderedVariables", [{"name": n, "expr": _expr(t)} for n, t in compiled["ordered_variables"]])
    if with_map:
        _put(d, "variables", {n: _expr(t) for n, t in compiled["ordered_variables"]})
    _put(d, "constants", compiled["constants"])


def _derived_role(dr):
    d = {"name": dr["name"], "parentRoles": {r: {} for r in dr["parent_roles"]}, "originFqn": dr["origin_fqn"]}
    _put(d, "condition", _cond(dr["condition"]))
    _vars(d, dr)
    return d


def _source_attrs(compiled, files):
    return {namer.policy_key_from_fqn(p["fqn"]): {"attributes": {"source": files[p["fqn"]]}} for p in compiled}


def _runnable(compiled, policies, files):
    first = compiled[0]
    out = {"fqn": first["fqn"], "compilerVersion": 2}
    if first["kind"] == "resource":
        pols = []
        for p in compiled:
            d = {"scopePermissions": _SP[p["scope_permissions"]]}
            _put(d, "scope", p["scope"])
            _put(d, "derivedRoles", {n: _derived_role(dr) for n, dr in p["derived_roles"].items()})
            rules = []
            for r in p["rules"]:
                rd = {"name": r["name"], "effect": "EFFECT_" + r["effect"]}
                _put(rd, "actions", {a: {} for a in r["actions"]})
                _put(rd, "roles", {a: {} for a in r["roles"]})
                _put(rd, "derivedRoles", {a: {} for a in r["derived_roles"]})
                _put(rd, "condition", _cond(r["condition"]))
                _put(rd, "emitOutput", _output(r["emit_output"]))
                rules.append(rd)
            _put(d, "rules", rules)
            _put(d, "schemas", policies[p["fqn"]]["resourcePolicy"].get("schemas"))
            _vars(d, p)
            pols.append(d)
        meta = {"fqn": first["fqn"], "resource": first["resource"], "version": first["version"], "sourceAttributes": _source_attrs(compiled, files)}
        body = {"meta": meta, "policies": pols}
        _put(body, "schemas", pols[-1].get("schemas"))   # compile.go:187-188: the root policy's
        out["resourcePolicy"] = body
    elif first["kind"] == "principal":
        pols = []
        for p in compiled:
            d = {"scopePermissions": _SP[p["scope_permissions"]]}
            _put(d, "scope", p["scope"])
            rr = {}
            for res, action_rules in p["resource_rules"].items():
                ars = []
                for a in action_rules:
                    ad = {"action": a["action"], "name": a["name"], "effect": "EFFECT_" + a["effect"]}
                    _put(ad, "condition", _cond(a["condition"]))
                    _put(ad, "emitOutput", _output(a["emit_output"]))
                    ars.append(ad)
                rr[res] = {"actionRules": ars}
            _put(d, "resourceRules", rr)
            _vars(d, p)
            pols.append(d)
        out["principalPolicy"] = {"meta": {"fqn": first["fqn"], "principal": first["principal"], "version": first["version"],
                                           "sourceAttributes": _source_attrs(compiled, files)}, "policies": pols}
    else:
        body = {"meta": {"fqn": first["fqn"], "version": first["version"], "sourceAttributes": _source_attrs(compiled, files)}, "role": first["role"]}
        _put(body, "scope", first["scope"])
        _put(body, "parentRoles", first["parent_roles"])
        res = {}
        for name, rules in first["resources"].items():
            rl = []
            for r in rules:
                rd = {"resource": r["resource"]}
                _put(rd, "name", r["name"])
                _put(rd, "allowActions", {a: {} for a in r["allow_actions"]})
                _put(rd, "condition", _cond(r["condition"]))
                _put(rd, "emitOutput", _output(r["emit_output"]))
                rl.append(rd)
            res[name] = {"rules": rl}
        _put(body, "resources", res)
        _vars(body, first, with_map=False)
        out["rolePolicy"] = body
    return out


def _numbers_as_floats(x):
    """protojson prints every google.protobuf.Value number as a double"""
    if isinstance(x, bool):
        return x
    if isinstance(x, (int, float)):
        return float(x)
    if isinstance(x, dict):
        return {k: _numbers_as_floats(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_numbers_as_floats(v) for v in x]
    return x


@pytest.mark.parametrize("case", [c for c in CASES if "golden" in c], ids=lambda c: c["name"])
def test_compiled_as_the_reference_compiled_it(case):
    policies, sources, files, main = _unit(case)
    compiled = pc.compile_unit(policies, main, sources)
    have = _runnable(compiled, policies, files)
    assert _numbers_as_floats(have) == _numbers_as_floats(case["golden"])
    for w in case["wantVariables"]:   # compile_test.go:76-78 requireVariables
        scope_policy = next(p for p in compiled if p["scope"] == w.get("scope", ""))
        if "derivedRoles" in w:
            for dr in w["derivedRoles"]:
                assert sorted(n for n, _ in scope_policy["derived_roles"][dr["name"]]["ordered_variables"]) == sorted(dr.get("variables") or [])
        assert sorted(n for n, _ in scope_policy["ordered_variables"]) == sorted(w.get("variables") or [])


def test_a_store_compiles_every_unit_and_gathers_every_error():
    """BatchCompile (compile.go:39-49): the errors of all the policies of a store in one report."""
    policies, sources = {}, {}
    for name in ("bad_variables", "unknown_derived_role"):
        case = next(c for c in CASES if c["name"] == name)
        p, s, _, _ = _unit(case)
        policies.update(p)
        sources.update(s)
    with pytest.raises(pc.CompileError) as got:
        pc.compile_all(policies, sources)
    kinds = {e.error for e in got.value.errors}
    assert {"undefined variable", "cyclical variable definitions", "invalid expression", "unknown derived role"} <= kinds
    assert str(got.value).startswith("%d compilation errors:\n" % len(got.value.errors))


def test_kinds_are_what_they_say():
    assert all(policy_kind(load_yaml_with_source(t, f)[0]) for c in CASES for f, t in c["files"].items())


def test_the_checks_reject_nothing_the_reference_accepts():
    """cel/check.py restates two of cel-go's checks: every expression of the reference's own known-answer sets (which its
    compiler accepted) must pass them."""
    from cerbos_amd.cel import check as celcheck
    from cerbos_amd.policy.compile import condition_exprs
    texts = [c["expr"] for c in load_json("cerbos_lib_kats.json")["cases"]]
    for case in load_json("cel_eval_cases.json"):
        texts += list(condition_exprs(pc.compile_condition({"match": case["condition"]})))
    assert len(texts) > 280
    for t in texts:
        ast, msgs = celcheck.compile_issues(t)
        assert ast is not None and msgs == [], (t, msgs)
    for store in ("store_policies.json",):
        policies = {policy_fqn(d): d for d in load_json(store)}
        assert pc.compile_all(policies, require_ancestors=True)
