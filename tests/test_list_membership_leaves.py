"""`x in <container the request brings>` as a classified fused leaf (celc.py _leaf_class 7: a string constant, 8: a column; the
kernels' flat_leaf decides it inline where round 3 filed an evaluation site for the interpreter's pre-pass): every shape the
operands can take, against oracle/check.py - lists with mixed element types, maps (membership among the KEYS), empty and long
containers, operands that are missing, null or of a type `in` has no overload for (an evaluation error, the rule does not
apply), needles that are not strings (the shared evaluator decides those).  Simulator and GPU."""
import pytest

from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000


def _expr(e):
    return {"match": {"expr": e}}


def _docs():
    rules = [
        {"actions": ["hold"], "roles": ["*"], "effect": "EFFECT_DENY", "condition": _expr('"legal-hold" in R.attr.labels')},
        {"actions": ["hold"], "roles": ["user"], "effect": "EFFECT_ALLOW"},
        {"actions": ["region"], "roles": ["user"], "effect": "EFFECT_ALLOW", "condition": _expr("R.attr.region in P.attr.regions")},
        {"actions": ["both"], "roles": ["user"], "effect": "EFFECT_ALLOW",
         "condition": {"match": {"all": {"of": [{"expr": '"a" in R.attr.labels'}, {"expr": "R.attr.region in P.attr.regions"}]}}}},
        {"actions": ["acl"], "roles": ["user"], "effect": "EFFECT_ALLOW", "condition": _expr("P.id in R.attr.acl")},
    ]
    # globs + a role policy keep the table off the flat kernels: cbh_walk2_kernel decides it (and its pre-pass what is left)
    rules.append({"actions": ["x:*"], "roles": ["user"], "effect": "EFFECT_ALLOW"})
    return [{"apiVersion": API, "resourcePolicy": {"resource": "doc", "version": "default", "rules": rules}}]


LABELS = [["legal-hold"], ["a", "legal-hold", "b"], ["a"], [], ["a", 1, True, None, ["legal-hold"]], {"legal-hold": 1}, {"x": "legal-hold"}, "legal-hold", 7, None,
          ["l%d" % i for i in range(70)] + ["legal-hold"], ["l%d" % i for i in range(70)], "__missing__"]
REGIONS = [["eu", "us"], ["us"], [], {"eu": True}, "eu", None, "__missing__", [1.0, "eu"], [["eu"]]]
REGION = ["eu", "apac", 1, 1.0, None, ["eu"], "__missing__"]


def _inputs():
    out = []
    for i, lab in enumerate(LABELS):
        for j, regs in enumerate(REGIONS):
            r = REGION[(i + j) % len(REGION)]
            rattr, pattr = {}, {}
            if lab != "__missing__":
                rattr["labels"] = lab
            if r != "__missing__":
                rattr["region"] = r
            if regs != "__missing__":
                pattr["regions"] = regs
            rattr["acl"] = [{"p1": 1}, ["p1", "p2"], {}, "p1"][(i + j) % 4]
            out.append({"requestId": "q%d_%d" % (i, j), "actions": ["hold", "region", "both", "acl"],
                        "principal": {"id": "p1", "roles": ["user"], "attr": pattr}, "resource": {"kind": "doc", "id": "d", "attr": rattr}})
    return out


def _run(make, close):
    rt = rule_table_from_policies(policies_from_docs(_docs()))
    lt = lower_rule_table(rt)
    assert lt.stats["gslots"][0] == 0, lt.stats["gslots"]     # no site of this table needs the interpreter
    ev = make(lt)
    orc = RuleTableOracle(rt)
    inputs = _inputs()
    try:
        outs, bad = ev.check(inputs, now_ns=NOW, allow_unsupported=True)
        assert not bad
        n_err = 0
        for inp, have in zip(inputs, outs):
            want = orc.check(inp, EvalParams(now_ns=NOW))
            assert norm_actions(have) == norm_actions(want), (inp, norm_actions(have), norm_actions(want))
            n_err += bool(want.get("evaluationErrors"))
        assert n_err > 20
        effects = {e["effect"] for o in outs for e in o["actions"].values()}
        assert effects == {"EFFECT_ALLOW", "EFFECT_DENY"}
    finally:
        if close:
            ev.close()


def test_membership_leaves_on_the_simulator():
    from test_hostsim_golden import HostSimEvaluator
    _run(lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.gpu
def test_membership_leaves_on_the_gpu():
    _run(lambda lt: HipEvaluator(lt, Conf()), True)
