"""Pins the oracle's CEL evaluator against the reference's condition KATs
(internal/engine/evaluator_test.go:22-48, internal/test/testdata/cel_eval/*.yaml,
frozen now = 2021-04-22T10:05:20.021-05:00)."""
import pytest

from cerbos_amd.policy.compile import compile_condition, condition_exprs
from helpers import CEL_EVAL_NOW_NS, load_json
from oracle import celeval
from oracle.check import EvalParams, _EvalContext

CASES = load_json("cel_eval_cases.json")

# Leaves that need cel-go features the oracle does not restate (parity-unpinned):
UNSUPPORTED = ("spiffe", "json.encode")


def _mk(case):
    inp = dict(case["request"])
    inp.setdefault("principal", {}).setdefault("id", "")
    inp.setdefault("resource", {}).setdefault("kind", "")
    if "auxData" not in inp and "aux_data" in inp:
        inp["auxData"] = inp["aux_data"]
    return _EvalContext(EvalParams(now_ns=CEL_EVAL_NOW_NS), inp)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_condition(case):
    cond = compile_condition({"match": case["condition"]})
    texts = list(condition_exprs(cond))
    if any(u in t for t in texts for u in UNSUPPORTED):
        pytest.skip("uses CEL extensions outside the restated subset")
    ev = _mk(case)
    assert ev.satisfies(cond, {}, {}) == case["want"]


def _leaves():
    out = []
    for case in CASES:
        if case["want"] is not True or "all" not in case["condition"]:
            continue
        for m in case["condition"]["all"]["of"]:
            if "expr" in m:
                out.append((case["name"], m["expr"], case))
    return out


@pytest.mark.parametrize("name,expr,case", _leaves(), ids=["%s-%d" % (n, i) for i, (n, _, _) in enumerate(_leaves())])
def test_leaf_true(name, expr, case):
    """Every leaf of an `all` condition whose golden result is true must itself be true."""
    if any(u in expr for u in UNSUPPORTED):
        pytest.skip("uses CEL extensions outside the restated subset")
    ev = _mk(case)
    assert celeval.evaluate(expr, ev._env({}, {})) is True, expr
