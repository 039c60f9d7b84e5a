"""The reference's service-level cases beyond check_resources/cr_case_* (tests/golden/service_more_cases.json, mined by
tools/make_golden_service.py from internal/test/testdata/server): CheckResourceSet, CheckResourceBatch, the CheckResources cases that
carry a JWT, and the playground's proxy / evaluate cases - the same three requests against the policies the request brings (a few
files of the store: partial stores the engine cases never see).  Each turned into the CheckInputs the service hands engine.Check
(cerbos_svc.go:147-166, 213-227, 274-287; playground_svc.go:131-240) with the effects, matched policies / scopes and effective
derived roles its response shows.  Three tiers: the oracle (pins it), the device source on the simulator, the MI355X."""
import json

import pytest

from helpers import assert_server_case, load_json, store_rule_table

CASES = load_json("service_more_cases.json")
NOW = 1_700_000_000_000_000_000


def _rule_table(case):
    if "policies" not in case:
        return store_rule_table()
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies
    return rule_table_from_policies(policies_from_docs(case["policies"]))


def _key(case):   # cases that share a store and its globals share an engine
    return json.dumps([case.get("policies"), case["globals"]], sort_keys=True)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle(case):
    from oracle.check import EvalParams, RuleTableOracle
    orc = RuleTableOracle(_rule_table(case))
    params = EvalParams(globals_=case["globals"], now_ns=NOW)
    assert assert_server_case(case, [orc.check(i, params) for i in case["inputs"]]) == len(case["inputs"])


def _run(make_evaluator, close=False):
    from cerbos_amd.engine import Conf
    from cerbos_amd.lower.blob import lower_rule_table
    evs, compared, flagged = {}, 0, 0
    for case in CASES:
        k = _key(case)
        if k not in evs:
            evs[k] = make_evaluator(lower_rule_table(_rule_table(case), case["globals"]), Conf(globals_=case["globals"]))
        outs, bad = evs[k].check(case["inputs"], now_ns=NOW, allow_unsupported=True)
        compared += assert_server_case(case, outs, skip=bad)
        flagged += len(bad)
    for ev in evs.values():
        if close:
            ev.close()
    return compared, flagged


def test_device_source_on_the_simulator():
    from test_hostsim_golden import HostSimEvaluator
    compared, flagged = _run(HostSimEvaluator)
    assert flagged == 0 and compared == sum(len(c["inputs"]) for c in CASES)


@pytest.mark.gpu
def test_gpu():
    from cerbos_amd.engine import HipEvaluator
    compared, flagged = _run(HipEvaluator, close=True)
    assert flagged == 0 and compared == sum(len(c["inputs"]) for c in CASES)


@pytest.mark.gpu
def test_gpu_bytes_in_bytes_out():
    """... and down the device road: serialized CheckInputs in, serialized CheckOutputs out."""
    from cerbos_amd import wire
    from cerbos_amd.engine import Conf, HipEvaluator
    from cerbos_amd.lower.blob import lower_rule_table
    evs, compared = {}, 0
    for case in CASES:
        k = _key(case)
        if k not in evs:
            evs[k] = HipEvaluator(lower_rule_table(_rule_table(case), case["globals"]), Conf(globals_=case["globals"]))
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in case["inputs"]])
        raw, flags = evs[k].check_pb(data, off, now_ns=NOW)
        assert not any(f & 1 for f in flags), case["name"]
        compared += assert_server_case(case, [wire.decode_check_output(r) for r in raw])
    for ev in evs.values():
        ev.close()
    assert compared == sum(len(c["inputs"]) for c in CASES)
