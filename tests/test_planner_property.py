"""What a plan MEANS, independent of the reference's fixtures: the filter PlanResources returns for (principal, action, kind), evaluated
on a concrete resource, is the decision CheckResources gives for that resource.  Generated stores (scopes and scope permissions, derived
roles, role policies with parent roles, principal policies, globs, the wide CEL pool) and requests; the plan is made without the
resource's attributes - and once more with half of them supplied, which must not change what it means; the filter is evaluated by the
oracle's CEL evaluator, the decision is the oracle's.  Cases where an evaluation meets a CEL error on either side are not compared (a
check absorbs an error per condition, a filter evaluated as one expression cannot) - they are counted, the rest must agree."""
import numpy as np
import pytest

from cerbos_amd.plan import Planner
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle import celeval
from oracle.check import EvalParams, RuleTableOracle, _EvalContext
from test_fuzz_parity import _policies, _requests

NOW = 1_700_000_000_000_000_000
_BIN = {"eq": "==", "ne": "!=", "lt": "<", "le": "<=", "gt": ">", "ge": ">=", "in": "in", "add": "+", "sub": "-", "mult": "*", "div": "/", "mod": "%"}
_GLOBAL = {"size", "timestamp", "duration", "now", "intersect", "hasIntersection", "isSubset", "except", "timeSince", "hierarchy", "string", "int", "double",
           "bool", "type", "dyn", "has", "inIPAddrRange"}
_MACROS = {"all", "exists", "exists_one", "map", "filter", "transformList", "transformMap", "transformMapEntry"}


class Unsupported(Exception):
    pass


def _lit(v):
    if v is None:
        return ("lit", "null", None)
    if isinstance(v, bool):
        return ("lit", "bool", v)
    if isinstance(v, (int, float)):
        return ("lit", "double", float(v))
    if isinstance(v, str):
        return ("lit", "string", v)
    if isinstance(v, list):
        return ("list", tuple(_lit(x) for x in v))
    return ("map", tuple((("lit", "string", k), _lit(x)) for k, x in v.items()))


def to_cel(op, bound=()):   # noqa: C901
    """A filter operand back into a CEL tree (the inverse of cerbos_amd/plan/filter.py build, as far as the generated stores need it)."""
    if "value" in op:
        return _lit(op["value"])
    if "variable" in op:
        parts = op["variable"].split(".")
        n = ("ident", parts[0])
        for f in parts[1:]:
            n = ("select", n, f)
        return n
    e = op["expression"]
    oper, ops = e["operator"], e["operands"]
    if oper in _MACROS:
        lam = ops[1]["expression"]
        assert lam["operator"] == "lambda"
        nv = 2 if oper.startswith("transform") else 1
        vars_ = tuple(o["variable"] for o in lam["operands"][-nv:])
        args = tuple(to_cel(o) for o in lam["operands"][:-nv])
        return ("comp", oper, to_cel(ops[0]), vars_, args)
    a = [to_cel(o) for o in ops]
    if oper in _BIN:
        return ("bin", _BIN[oper], a[0], a[1])
    if oper in ("and", "or"):
        out = a[0]
        for x in a[1:]:
            out = (oper, out, x)
        return out
    if oper == "not":
        return ("not", a[0])
    if oper == "if":
        return ("tern", a[0], a[1], a[2])
    if oper == "index":
        return ("index", a[0], a[1])
    if oper == "list":
        return ("list", tuple(a))
    if oper == "get-field":
        return ("select", a[0], ops[1]["variable"])
    if oper in ("struct", "set-field", "-_"):
        raise Unsupported(oper)
    if oper in _GLOBAL and not (oper in ("timeSince", "inIPAddrRange", "except") and len(a) >= 1 and oper != "intersect"):
        return ("call", oper, None, tuple(a))
    if not a:
        return ("call", oper, None, ())
    return ("call", oper, a[0], tuple(a[1:]))


def holds(flt, ctx):
    """-> True / False, or None when evaluating the filter is a CEL error"""
    if flt["kind"] != "KIND_CONDITIONAL":
        return flt["kind"] == "KIND_ALWAYS_ALLOWED"
    try:
        return celeval.evaluate(to_cel(flt["condition"]), ctx._env({}, {})) is True
    except celeval.CelError:
        return None


@pytest.mark.parametrize("seed", range(24))
def test_a_plan_s_filter_on_a_resource_is_the_check_s_decision(seed):
    rng = np.random.default_rng(500 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    planner, oracle = Planner(rt), RuleTableOracle(rt)
    compared = skipped = 0
    for lenient in (False, True):
        params = EvalParams(now_ns=NOW, lenient_scope_search=lenient)
        for inp in _requests(rng, 120):
            # ONE role: with several the reference's plan is deliberately not the check's fold - it ANDs NOT(deny of any role) onto the
            # OR of the roles' allows (plan.go:330-377), where a check lets one role's ALLOW stand against another role's DENY
            # (check.go:429-442); the goldens pin that behaviour, this property pins what a plan means where the two agree
            inp = dict(inp, principal=dict(inp["principal"], roles=list(inp["principal"]["roles"][:1])))
            if not inp["principal"]["roles"]:
                continue
            res = inp["resource"]
            out = oracle.check(inp, params)
            ctx = _EvalContext(params, inp)
            attrs = res.get("attr") or {}
            half = {k: v for n, (k, v) in enumerate(sorted(attrs.items())) if n % 2 == 0}
            for action in inp["actions"]:
                want = out["actions"][action]["effect"] == "EFFECT_ALLOW"
                for known in ({}, half):
                    pin = {"principal": inp["principal"], "actions": [action], "auxData": inp.get("auxData"),
                           "resource": {"kind": res["kind"], "policyVersion": res.get("policyVersion", ""), "scope": res.get("scope", ""), "attr": known}}
                    plan = planner.plan(pin, lenient_scope_search=lenient, now_ns=NOW)
                    try:
                        have = holds(plan["filter"], ctx)
                    except Unsupported:
                        have = None
                    # (a principal policy's unconditional DENY becomes a FALSE allow OR-ed with the resource policies' plan, plan.go:330-343,
                    #  358-365 - where a check lets the principal policy decide first, check.go:445-448: not compared either)
                    # ... nor a chain with a REQUIRE_PARENTAL_CONSENT scope: the plan ANDs the child's allow with the parent's
                    # (plan.go:266-277) also where the child's condition fails and a check lets the parent allow alone (check.go:416-425)
                    chain = [res.get("scope", "")] + list(__import__("cerbos_amd.namer", fromlist=["x"]).scope_parents(res.get("scope", "")))
                    if any(rt["scope_permissions"].get(sc) == 2 for sc in chain):
                        skipped += 1
                        continue
                    if any(k.startswith("principal.") for k in plan["effectivePolicies"]):
                        skipped += 1
                        continue
                    if have is None or out["evaluationErrors"] or plan["evaluationErrors"]:
                        skipped += 1
                        continue
                    compared += 1
                    assert have == want, (seed, lenient, action, bool(known), plan["filterDebug"], inp)
    assert compared > 150, (compared, skipped)
