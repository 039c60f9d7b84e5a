"""The known-answer tests the reference keeps in Go on the hot path itself - TestCELErrorsCheck
(internal/ruletable/cel_errors_test.go:277-356) and TestStrictEvaluationCheck
(internal/ruletable/strict_evaluation_test.go:17-92) over the six policies of newCELErrorsHarness - mined into
tests/golden/ruletable_cel_errors.json by tools/make_golden_ruletable_tests.py: CEL runtime errors fail open
(the erroring rule is skipped) and are reported; under EvalParams.StrictEvaluation they deny exactly the actions
whose evaluation met the error (a variable, a derived-role definition, runtime.effectiveDerivedRoles).

* the oracle: every asserted effect, policy and the ORDERED expression list of CheckOutput.evaluation_errors;
* the device path (kernel source on the host simulator; GPU tier: the kernel): every asserted effect and policy, and
  "an error was absorbed" (CBH_ST_CEL_ERROR on some action) exactly where the reference reports errors."""
import pytest

from cerbos_amd import capi
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json
from oracle.check import EvalParams, RuleTableOracle

FIX = load_json("ruletable_cel_errors.json")
NOW = 1_700_000_000_000_000_000


def _input(case):
    attr = {}
    if "number" in case["amount"]:
        attr["amount"] = case["amount"]["number"]
    elif "string" in case["amount"]:
        attr["amount"] = case["amount"]["string"]
    return {"requestId": FIX["requestId"], "actions": case["actions"], "principal": dict(FIX["principal"]),
            "resource": {"kind": case["kind"], "id": FIX["resourceId"], "attr": attr}}


@pytest.fixture(scope="module")
def table():
    rt = rule_table_from_policies(policies_from_docs(FIX["policies"]))
    return rt, lower_rule_table(rt)


@pytest.mark.parametrize("case", FIX["cases"], ids=[("strict-" if c["strict"] else "") + c["name"] for c in FIX["cases"]])
def test_oracle_matches_the_reference_assertions(table, case):
    rt, _ = table
    out = RuleTableOracle(rt).check(_input(case), EvalParams(now_ns=NOW, strict_evaluation=case["strict"]))
    for a, eff in case["wantEffects"].items():
        assert out["actions"][a]["effect"] == eff, (case["name"], a)
    for a, pol in case["wantPolicies"].items():
        assert out["actions"][a]["policy"] == pol, (case["name"], a)
    exprs = [e["celError"]["expression"] for e in out.get("evaluationErrors") or []]
    assert exprs == case["wantErrorExpressions"], case["name"]
    assert all(e["celError"]["message"] for e in out.get("evaluationErrors") or [])


def _device(table, make_evaluator, close):
    rt, lt = table
    assert not lt.unsupported, lt.unsupported
    ev = make_evaluator(lt)
    try:
        for strict in (False, True):
            cases = [c for c in FIX["cases"] if c["strict"] == strict]
            inputs = [_input(c) for c in cases]
            batch = Flattener(lt).flatten(inputs)
            res = ev.table.check(batch, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES | (capi.F_STRICT_EVALUATION if strict else 0))
            outs, bad = ev.assemble(inputs, batch, res, "default", allow_unsupported=True)
            assert not bad
            t = 0
            for c, inp, out in zip(cases, inputs, outs):
                for a, eff in c["wantEffects"].items():
                    assert out["actions"][a]["effect"] == eff, (c["name"], strict, a)
                for a, pol in c["wantPolicies"].items():
                    assert out["actions"][a]["policy"] == pol, (c["name"], strict, a)
                na = len(inp["actions"])
                absorbed = bool((res.status[t:t + na] == capi.ST_CEL_ERROR).any())
                assert absorbed == bool(c["wantErrorExpressions"]), (c["name"], strict)
                t += na
    finally:
        if close:
            ev.close()


def test_kernel_source_matches_the_reference_assertions(table):
    from test_hostsim_golden import HostSimEvaluator
    _device(table, lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.gpu
def test_gpu_matches_the_reference_assertions(table):
    _device(table, lambda lt: HipEvaluator(lt, Conf()), True)
