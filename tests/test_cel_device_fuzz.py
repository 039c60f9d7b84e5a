"""Generated CEL expressions over REQUEST values on the device path (lowering + kernel source on the host simulator) against
the oracle's evaluator: each expression is the condition of its own ALLOW rule, so a request's answer for action k is the
truth of expression k - ALLOW, DENY (false or a CEL error) - or the input is flagged UNSUPPORTED; never another answer.

The generator is test_cel_fold's with request attributes in the place of some literals: typed right, typed wrong, missing."""
import random

import pytest

from cerbos_amd import capi
from cerbos_amd.engine import Conf
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle import celeval
from oracle.check import EvalParams, _EvalContext
from test_cel_fold import _gen

NOW = 1_700_000_000_000_000_000
API = "api.cerbos.dev/v1"
PER_STORE = 40
ATTRS = {
    "int": ["R.attr.i1", "P.attr.i2", "R.attr.missing", "R.attr.s1"],
    "dbl": ["R.attr.d1", "P.attr.d2", "R.attr.i1"],
    "str": ["R.attr.s1", "P.attr.s2", "R.id", "P.id", "R.kind", "R.attr.missing", "R.attr.i1"],
    "bool": ["R.attr.b1", "P.attr.b2", "R.attr.missing"],
    "ilist": ["R.attr.il", "P.attr.il2", "R.attr.sl"],
    "slist": ["R.attr.sl", "P.attr.sl2", "P.roles", "R.attr.missing"],
}
STRS = ["", "a", "abc", "a.b.c", "a,b,,c", "  pad  ", "Ünï", "x.y", "ABC", "a.b"]


class _Rng(random.Random):
    """test_cel_fold._gen draws its leaves with rng.choice(list): hand it attribute paths now and then."""
    swap = 0.35


def _expr(rng, want):
    text = _gen(rng, rng.randrange(1, 4), want)
    # replace some literals of the generated text by request values of the same type
    out, i = [], 0
    import re
    tokens = re.split(r'("(?:[^"\\]|\\.)*"|\b\d+\.\d+\b|\b\d+\b|\btrue\b|\bfalse\b)', text)
    for t in tokens:
        if t and rng.random() < 0.3:
            if t[0] == '"' and 'split(' not in "".join(out[-1:]) and not "".join(out).endswith(("split(", "hierarchy(", ", ")):
                t = rng.choice(ATTRS["str"])
            elif re.fullmatch(r"\d+\.\d+", t):
                t = rng.choice(ATTRS["dbl"])
            elif re.fullmatch(r"\d+", t) and not "".join(out).endswith(("[", "substring(", "charAt(", "range(", "(x, x * ")):
                t = rng.choice(ATTRS["int"])
            elif t in ("true", "false"):
                t = rng.choice(ATTRS["bool"])
        out.append(t)
    return "".join(out)


def _value(rng, kind):
    r = rng.random()
    if kind == "int":
        return rng.choice([0, 1, 2, 3, 7, -1, 42, 100])
    if kind == "dbl":
        return rng.choice([0.5, 1.5, -3.25, 1000.0, 2.0])
    if kind == "str":
        return rng.choice(STRS)
    if kind == "bool":
        return r < 0.5
    if kind == "ilist":
        return [rng.choice([0, 1, 2, 3, 7]) for _ in range(rng.randrange(0, 4))]
    return [rng.choice(STRS) for _ in range(rng.randrange(0, 4))]


def _request(rng, n_actions):
    def attrs(spec):
        out = {}
        for name, kind in spec:
            if rng.random() < 0.1:
                continue                                   # missing
            out[name] = _value(rng, kind if rng.random() > 0.06 else rng.choice(["int", "str", "bool", "slist"]))   # sometimes the wrong type
        return out
    return {"requestId": "x", "principal": {"id": rng.choice(["p1", "abc", "a.b"]), "roles": rng.sample(["user", "admin", "abc", "a"], rng.randrange(1, 3)),
                                           "attr": attrs([("i2", "int"), ("d2", "dbl"), ("s2", "str"), ("b2", "bool"), ("il2", "ilist"), ("sl2", "slist")])},
            "resource": {"kind": "kat", "id": rng.choice(["r1", "abc", "x.y"]),
                         "attr": attrs([("i1", "int"), ("d1", "dbl"), ("s1", "str"), ("b1", "bool"), ("il", "ilist"), ("sl", "slist")])},
            "actions": ["a%d" % k for k in range(n_actions)]}


def _truth(expr, inp):
    ev = _EvalContext(EvalParams(now_ns=NOW), inp)
    try:
        return celeval.evaluate(expr, ev._env({}, {})) is True
    except celeval.CelError:
        return False


def run_seed(seed, make=None, n_requests=12):
    from test_hostsim_golden import HostSimEvaluator
    rng = _Rng(77_000 + seed)
    exprs = []
    while len(exprs) < PER_STORE:
        e = _expr(rng, rng.choice(["bool", "bool", "bool"]))
        try:
            from cerbos_amd.cel import parser
            parser.parse(e)
        except Exception:   # noqa: BLE001 - the substitution can break a literal-only construct
            continue
        exprs.append(e)
    rules = [{"actions": ["a%d" % k], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": e}}} for k, e in enumerate(exprs)]
    rt = rule_table_from_policies(policies_from_docs([{"apiVersion": API, "resourcePolicy": {"resource": "kat", "version": "default", "rules": rules}}]))
    lt = lower_rule_table(rt)
    ev = (make or (lambda t: HostSimEvaluator(t, Conf())))(lt)
    decided = flagged = 0
    for _ in range(n_requests):
        inp = _request(rng, len(exprs))
        res = ev.table.check(Flattener(lt).flatten([inp]), now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        for k, e in enumerate(exprs):
            st, eff = int(res.status[k]), int(res.effect[k])
            if st == capi.ST_UNSUPPORTED:
                flagged += 1
                continue
            decided += 1
            want = _truth(e, inp)
            assert (eff == 1) == want, (seed, e, inp, st, eff, want)
    return decided, flagged


@pytest.mark.parametrize("seed", range(12))
def test_generated_expressions_over_request_values(seed):
    decided, flagged = run_seed(seed)
    assert decided > 100, (decided, flagged)
