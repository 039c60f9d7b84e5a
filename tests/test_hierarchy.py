"""cerbos.lib.hierarchy predicates on the device path (internal/conditions/types/hierarchy.go:259-385) against the
oracle's restatement over random dot-delimited paths: empty strings, empty segments ("a..b", ".a", "a."), equal paths,
prefixes that are not segment prefixes ("a.bc" vs "a.b").  hierarchy(list), missing attributes and non-strings must
come out as the reference has them (UNSUPPORTED / evaluation error), never as an answer.
CPU tier: the kernel source on the host simulator; GPU tier: the kernel."""
import numpy as np
import pytest

from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000
PREDICATES = ["ancestorOf", "descendentOf", "immediateParentOf", "immediateChildOf", "siblingOf", "overlaps"]
CONDS = {p: "hierarchy(R.attr.a).%s(hierarchy(P.attr.b))" % p for p in PREDICATES}
CONDS["const"] = 'hierarchy("acme.hr").ancestorOf(hierarchy(R.attr.a))'
CONDS["negated"] = "!hierarchy(R.attr.a).overlaps(hierarchy(P.attr.b))"


def _docs(conds):
    return [{"apiVersion": API, "resourcePolicy": {"resource": "org", "version": "default", "rules": [
        {"actions": [n], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": e}}} for n, e in conds.items()]}}]


def _path(rng):
    segs = ["acme", "hr", "uk", "a", "b", "bc", ""]
    n = int(rng.integers(0, 5))
    return ".".join(str(rng.choice(segs)) for _ in range(n))


def _pairs(rng, n):
    out = []
    for _ in range(n):
        a = _path(rng)
        r = rng.random()
        if r < 0.3:
            b = a
        elif r < 0.55:
            b = a + "." + _path(rng)
        elif r < 0.7:
            b = a + str(rng.choice(["c", ".", ""]))
        elif r < 0.8 and "." in a:
            b = a.rsplit(".", 1)[0] + "." + str(rng.choice(["x", "hr", ""]))
        else:
            b = _path(rng)
        out.append((a, b) if rng.random() < 0.5 else (b, a))
    return out


def _run(make_evaluator, close):
    rng = np.random.default_rng(2024)
    rt = rule_table_from_policies(policies_from_docs(_docs(CONDS)))
    lt = lower_rule_table(rt)
    assert not lt.unsupported, lt.unsupported
    inputs = [{"requestId": "q%d" % i, "actions": list(CONDS), "principal": {"id": "p", "roles": ["user"], "attr": {"b": b}},
               "resource": {"kind": "org", "id": "r%d" % i, "attr": {"a": a}}} for i, (a, b) in enumerate(_pairs(rng, 400))]
    inputs.append(dict(inputs[0], requestId="missing", resource={"kind": "org", "id": "m", "attr": {}}))          # evaluation error
    inputs.append(dict(inputs[0], requestId="number", resource={"kind": "org", "id": "n", "attr": {"a": 3.0}}))   # no such overload
    ev = make_evaluator(lt)
    try:
        outs, bad = ev.check(inputs, now_ns=NOW, allow_unsupported=True)
        # hierarchy(list) is valid CEL outside the device subset: flagged
        _, bad_list = ev.check([dict(inputs[0], resource={"kind": "org", "id": "l", "attr": {"a": ["acme", "hr"]}})], now_ns=NOW, allow_unsupported=True)
    finally:
        if close:
            ev.close()
    assert not bad and bad_list == [0]
    orc = RuleTableOracle(rt)
    allowed = dict.fromkeys(CONDS, 0)
    for inp, have in zip(inputs, outs):
        want = orc.check(inp, EvalParams(now_ns=NOW))
        assert norm_actions(have) == norm_actions(want), (inp["resource"]["attr"], inp["principal"]["attr"], have["actions"], want["actions"])
        for a, e in want["actions"].items():
            allowed[a] += e["effect"] == "EFFECT_ALLOW"
    assert all(2 < v < len(inputs) - 10 for v in allowed.values()), allowed   # every predicate both holds and fails in the sample


def test_hierarchy_kernel_source_vs_oracle():
    from test_hostsim_golden import HostSimEvaluator
    _run(lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.gpu
def test_hierarchy_on_gpu():
    _run(lambda lt: HipEvaluator(lt, Conf()), True)
