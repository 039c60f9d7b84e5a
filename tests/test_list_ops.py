"""intersect() iterates the SHORTER of its two lists (the reference swaps the operands, internal/conditions/cerbos_lib.go:434-437):
element order and duplicates of the result follow that list.  Expected values below are worked out from that source by hand;
the oracle and the device (host simulator in the CPU tier, the kernel in the GPU tier) must both give them."""
import pytest

from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
CONDS = {"eq": "intersect(R.attr.a, R.attr.b) == P.attr.want", "size": "intersect(R.attr.a, R.attr.b).size() == P.attr.n",
         "first": "intersect(R.attr.a, R.attr.b)[0] == P.attr.first"}
# (a, b, intersect(a, b) in the reference)
CASES = [
    ([1, 1, 2], [1, 2], [1, 2]),                 # a is longer: b is iterated - no duplicate
    ([1, 2], [1, 1, 2], [1, 2]),
    (["b", "a", "c"], ["a", "b"], ["a", "b"]),   # order of the shorter list
    (["a", "b"], ["b", "a", "c"], ["a", "b"]),
    ([1, 1], [1, 2], [1, 1]),                    # equal lengths: the first list is iterated
    (["x", "y", "y"], ["y", "y", "y", "q"], ["y", "y"]),
    (["q"], ["x", "y"], []),
]


def _run(make_evaluator, close):
    docs = [{"apiVersion": API, "resourcePolicy": {"resource": "lists", "version": "default", "rules": [
        {"actions": [n], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": e}}} for n, e in CONDS.items()]}}]
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    assert not lt.unsupported, lt.unsupported
    inputs, expect = [], []
    for i, (a, b, want) in enumerate(CASES):
        for wrong in (False, True):
            w = (want + want[:1] if want else ["zz"]) if wrong else want
            inputs.append({"requestId": "q%d%d" % (i, wrong), "actions": list(CONDS), "resource": {"kind": "lists", "id": "r", "attr": {"a": a, "b": b}},
                           "principal": {"id": "p", "roles": ["user"], "attr": {"want": w, "n": len(w), "first": w[0] if w else "none"}}})
            expect.append({"eq": not wrong, "size": not wrong, "first": bool(want) and (not wrong or w[0] == want[0])})
    ev = make_evaluator(lt)
    try:
        outs, bad = ev.check(inputs, now_ns=0, allow_unsupported=True)
    finally:
        if close:
            ev.close()
    assert not bad
    orc = RuleTableOracle(rt)
    for inp, have, exp in zip(inputs, outs, expect):
        want = orc.check(inp, EvalParams(now_ns=0))
        for name, allowed in exp.items():
            attrs = (inp["resource"]["attr"], inp["principal"]["attr"], name)
            assert (want["actions"][name]["effect"] == "EFFECT_ALLOW") == allowed, ("oracle",) + attrs
            assert (have["actions"][name]["effect"] == "EFFECT_ALLOW") == allowed, ("device",) + attrs


def test_intersect_order_kernel_source_and_oracle():
    from test_hostsim_golden import HostSimEvaluator
    _run(lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.gpu
def test_intersect_order_on_gpu():
    _run(lambda lt: HipEvaluator(lt, Conf()), True)
