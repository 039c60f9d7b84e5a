"""String functions the device evaluates without building strings - indexOf / lastIndexOf (code-point indices) and
equality through lowerAscii / upperAscii (cel-go ext/strings.go) - against oracle/celeval.py over random strings with
non-ASCII characters, repeated and overlapping needles, empty strings, non-string and missing operands.
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
CONDS = {
    "first": 'R.attr.s.indexOf(P.attr.n) == 1', "first_none": 'R.attr.s.indexOf(P.attr.n) == -1', "first_ge": 'R.attr.s.indexOf("a") >= 2',
    "last": 'R.attr.s.lastIndexOf(P.attr.n) >= 3', "last_eq_first": 'R.attr.s.lastIndexOf(P.attr.n) == R.attr.s.indexOf(P.attr.n)',
    "empty_needle": 'R.attr.s.lastIndexOf("") == size(R.attr.s)',
    "lower": 'R.attr.s.lowerAscii() == P.attr.n', "upper": 'R.attr.s.upperAscii() == "ABÉ"', "lit_lower": '"ABA".lowerAscii() == R.attr.s',
    "both": 'R.attr.s.lowerAscii() == P.attr.n.lowerAscii()', "mixed": 'R.attr.s.upperAscii() != P.attr.n.lowerAscii()',
    "ne": 'P.attr.n != R.attr.s.lowerAscii()',
}


def _docs():
    return [{"apiVersion": API, "resourcePolicy": {"resource": "text", "version": "default", "rules": [
        {"actions": [n], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": e}}} for n, e in CONDS.items()]}}]


def _word(rng, alphabet, lo, hi):
    return "".join(str(rng.choice(alphabet)) for _ in range(int(rng.integers(lo, hi))))


def _run(make_evaluator, close):
    rng = np.random.default_rng(99)
    rt = rule_table_from_policies(policies_from_docs(_docs()))
    lt = lower_rule_table(rt)
    assert not lt.unsupported, lt.unsupported
    alphabet = list("abAB") + ["é", "É", "日"]
    inputs = []
    for i in range(400):
        s = _word(rng, alphabet, 0, 7)
        r = rng.random()
        n = s.lower() if r < 0.2 else (s[int(rng.integers(0, len(s) + 1)):][:int(rng.integers(0, 3))] if r < 0.6 else _word(rng, alphabet, 0, 3))
        inputs.append({"requestId": "q%d" % i, "actions": list(CONDS), "principal": {"id": "p", "roles": ["user"], "attr": {"n": n}},
                       "resource": {"kind": "text", "id": "r%d" % i, "attr": {"s": s}}})
    for k, s in enumerate(("abé", "ABÉ", "aba", "ABA", "AbA")):   # values the constant comparisons are about
        inputs.append(dict(inputs[0], requestId="fixed%d" % k, resource={"kind": "text", "id": "f%d" % k, "attr": {"s": s}}))
    inputs.append(dict(inputs[0], requestId="num", resource={"kind": "text", "id": "x", "attr": {"s": 7.0}}))
    inputs.append(dict(inputs[0], requestId="missing", resource={"kind": "text", "id": "y", "attr": {}}))
    inputs.append(dict(inputs[0], requestId="n_num", principal={"id": "p", "roles": ["user"], "attr": {"n": True}}))
    ev = make_evaluator(lt)
    try:
        outs, bad = ev.check(inputs, now_ns=NOW, allow_unsupported=True)
    finally:
        if close:
            ev.close()
    assert not bad
    orc = RuleTableOracle(rt)
    allowed = dict.fromkeys(CONDS, 0)
    for inp, have in zip(inputs, outs):
        want = orc.check(inp, EvalParams(now_ns=NOW))
        assert norm_actions(have) == norm_actions(want), (inp["resource"]["attr"], inp["principal"]["attr"], have["actions"], want["actions"])
        for a, e in want["actions"].items():
            allowed[a] += e["effect"] == "EFFECT_ALLOW"
    assert sum(v > 0 for v in allowed.values()) >= len(CONDS) - 2, allowed   # the conditions discriminate


def test_string_functions_kernel_source_vs_oracle():
    from test_hostsim_golden import HostSimEvaluator
    _run(lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.gpu
def test_string_functions_on_gpu():
    _run(lambda lt: HipEvaluator(lt, Conf()), True)
