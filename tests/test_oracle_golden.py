"""Pins the CPU oracle against the reference's own golden fixtures
(internal/engine/engine_test.go:46-210 TestCheck / TestCheckWithLenientScopeSearch)."""
import pytest

from helpers import load_json, norm_actions, store_rule_table
from oracle.check import EvalParams, RuleTableOracle

CASES = load_json("engine_cases.json")


@pytest.fixture(scope="module")
def oracle():
    return RuleTableOracle(store_rule_table())


def _modes(case):
    return [False, True] if case["lenient"] is None else [case["lenient"]]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_engine_case(oracle, case):
    for lenient in _modes(case):
        params = EvalParams(globals_={"environment": "test"}, now_ns=1_700_000_000_000_000_000,
                            lenient_scope_search=lenient)
        for inp, want in zip(case["inputs"], case["wantOutputs"]):
            have = oracle.check(inp, params)
            assert norm_actions(have) == norm_actions(want), (case["name"], lenient)
            assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or [])
            assert have["requestId"] == want.get("requestId", "")
            assert have["resourceId"] == want.get("resourceId", "")
            want_errs = want.get("evaluationErrors") or []
            assert have["evaluationErrors"] == want_errs
            want_outs = sorted(want.get("outputs") or [], key=lambda o: o["src"])
            have_outs = sorted(have["outputs"], key=lambda o: o["src"])
            if "output_now" not in case["name"]:
                assert have_outs == want_outs
        # the call's AuditTrail.EffectivePolicies (engine.go:289-338 merges the inputs' trails; check.go:302-304): the keys of the
        # case's decision log
        if case["hasDecisionLogs"] and not case["wantError"]:
            touched = set()
            for inp in case["inputs"]:
                touched.update(oracle.check(inp, params)["effectivePolicies"])
            assert sorted(touched) == case["wantEffectivePolicies"], (case["name"], lenient)


SERVER_CASES = load_json("server_check_cases.json")


@pytest.mark.parametrize("case", SERVER_CASES, ids=[c["name"] for c in SERVER_CASES])
def test_service_level_check_resources_case(oracle, case):
    """internal/test/testdata/server/checks/check_resources: the same store seen through CheckResources."""
    from helpers import assert_server_case
    params = EvalParams(globals_={"environment": "test"}, now_ns=1_700_000_000_000_000_000)
    assert assert_server_case(case, [oracle.check(i, params) for i in case["inputs"]]) == len(case["inputs"])


VERIFY_VECTORS = load_json("verify_vectors.json")


def test_policy_test_framework_vectors(oracle):
    """Effects the reference engine itself returned while running the policy-test-framework fixtures
    (testdata/verify/cases/*.golden, mined by tools/make_golden_verify.py): test-level `now`, JWT claims,
    globals, default policy version and lenient scope search on the golden store."""
    from helpers import rfc3339_ns
    n = 0
    for v in VERIFY_VECTORS:
        params = EvalParams(globals_=v["globals"], now_ns=rfc3339_ns(v["now"]) if v["now"] else 1_700_000_000_000_000_000,
                            default_policy_version=v["defaultPolicyVersion"], default_scope=v["defaultScope"],
                            lenient_scope_search=v["lenient"], strict_evaluation=v["strict"])
        have = oracle.check(v["input"], params)
        assert {a: e["effect"] for a, e in have["actions"].items()} == v["want"], (v["suite"], v["test"])
        n += len(v["want"])
    assert n > 100
