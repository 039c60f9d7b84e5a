"""GPU tier: the reference's golden engine cases through the real library on an MI355X
(lowering -> cbh_table_load -> flatten -> cbh_check_batch -> response assembly)."""
import pytest

from cerbos_amd.engine import Conf, HipEvaluator
from helpers import load_json, norm_actions, store_rule_table

pytestmark = pytest.mark.gpu

CASES = load_json("engine_cases.json")
GLOBALS = {"environment": "test"}
EXPECT_UNSUPPORTED = set()  # see tests/test_hostsim_golden.py


@pytest.fixture(scope="module")
def evaluator():
    ev = HipEvaluator.from_rule_table(store_rule_table(), Conf(globals_=GLOBALS))
    yield ev
    ev.close()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_engine_case(evaluator, case):
    modes = [False, True] if case["lenient"] is None else [case["lenient"]]
    for lenient in modes:
        outs, bad = evaluator.check(case["inputs"], now_ns=1_700_000_000_000_000_000,
                                    lenient_scope_search=lenient, allow_unsupported=True)
        for i, (have, want) in enumerate(zip(outs, case["wantOutputs"])):
            if i in bad:
                assert case["name"] in EXPECT_UNSUPPORTED
                continue
            assert norm_actions(have) == norm_actions(want), (case["name"], lenient)
            assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or [])


def test_engine_cases_through_the_native_ingest():
    """The same cases entering as serialized CheckInput bytes: wire.py -> libcerbos_ingest.so -> cbh_check_batch."""
    from cerbos_amd.lower.blob import lower_rule_table
    ev = HipEvaluator(lower_rule_table(store_rule_table(), GLOBALS), Conf(globals_=GLOBALS), native_ingest=True)
    compared = 0
    for case in CASES:
        for lenient in ([False, True] if case["lenient"] is None else [case["lenient"]]):
            outs, bad = ev.check(case["inputs"], now_ns=1_700_000_000_000_000_000, lenient_scope_search=lenient, allow_unsupported=True)
            for i, (have, want) in enumerate(zip(outs, case["wantOutputs"])):
                if i in bad:
                    assert case["name"] in EXPECT_UNSUPPORTED
                    continue
                assert norm_actions(have) == norm_actions(want), (case["name"], lenient)
                compared += 1
    ev.close()
    assert compared > 60


def test_engine_cases_bytes_in_bytes_out():
    """check_pb on the GPU: serialized CheckInput -> libcerbos_ingest.so -> cbh_check_batch -> serialized CheckOutput."""
    from cerbos_amd import wire
    from cerbos_amd.lower.blob import lower_rule_table
    ev = HipEvaluator(lower_rule_table(store_rule_table(), GLOBALS), Conf(globals_=GLOBALS))
    compared = 0
    for case in CASES:
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in case["inputs"]])
        for lenient in ([False, True] if case["lenient"] is None else [case["lenient"]]):
            raw, flags = ev.check_pb(data, off, now_ns=1_700_000_000_000_000_000, lenient_scope_search=lenient)
            for have, want, f in zip(raw, case["wantOutputs"], flags):
                if f & 1:
                    assert case["name"] in EXPECT_UNSUPPORTED
                    continue
                have = wire.decode_check_output(have)
                assert norm_actions(have) == norm_actions(want), (case["name"], lenient)
                assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or [])
                compared += 1
    ev.close()
    assert compared > 60


SERVER_CASES = load_json("server_check_cases.json")


@pytest.mark.parametrize("case", SERVER_CASES, ids=[c["name"] for c in SERVER_CASES])
def test_service_level_check_resources_case(evaluator, case):
    """The reference's service-level CheckResources cases (server/checks/check_resources) on the GPU."""
    from helpers import assert_server_case
    outs, bad = evaluator.check(case["inputs"], now_ns=1_700_000_000_000_000_000, allow_unsupported=True)
    assert assert_server_case(case, outs, skip=bad) + len(bad) == len(case["inputs"])
    assert len(bad) < len(case["inputs"])


def test_policy_test_framework_vectors():
    """tests/golden/verify_vectors.json - effects the reference engine returned for its policy-test-framework
    fixtures (test-level now, JWT claims, named JWTs, globals, strict / lenient modes) - on the GPU."""
    from cerbos_amd.lower.blob import lower_rule_table
    from test_hostsim_golden import _run_verify_vectors
    evs = []

    def make(globals_):
        evs.append(HipEvaluator(lower_rule_table(store_rule_table(), globals_), Conf(globals_=globals_)))
        return evs[-1]
    assert _run_verify_vectors(make) == (len(load_json("verify_vectors.json")), 0)
    for ev in evs:
        ev.close()
