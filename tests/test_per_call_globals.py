"""evaluator.EvalParams.Globals per CALL (internal/evaluator/evaluator.go:52-57, 98-106): a table lowered with
``per_call_globals`` reads `G.x` from the globals each call brings - as attribute columns of root "G" that every flattener
fills (flatten.py, cbi_flatten_pb_g, cbh_wire_flatten) - so ONE image answers the policy-test framework's vector groups,
whose globals differ (round 2 lowered the store once per globals set)."""
import json

import numpy as np
import pytest

from cerbos_amd import wire
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from helpers import load_json, rfc3339_ns, store_rule_table
from test_hostsim_golden import HostSimEvaluator

VECTORS = load_json("verify_vectors.json")
EFFECT = {"EFFECT_ALLOW": 1, "EFFECT_DENY": 2}


def _one_image(make):
    """every vector through ONE evaluator, its group's globals passed with the call; dict path and bytes path"""
    ev = make()
    assert ev.lt.per_call_globals and any(root == "G" for root, _ in ev.lt.columns)
    groups = {json.dumps(v["globals"], sort_keys=True) for v in VECTORS}
    assert len(groups) >= 2   # the fixture really has different globals
    compared = 0
    for v in VECTORS:
        now = rfc3339_ns(v["now"]) if v["now"] else 1_700_000_000_000_000_000
        kw = dict(now_ns=now, lenient_scope_search=v["lenient"], strict_evaluation=v["strict"],
                  default_policy_version=v["defaultPolicyVersion"], default_scope=v["defaultScope"], globals_=v["globals"])
        outs, bad = ev.check([v["input"]], allow_unsupported=True, **kw)
        assert not bad
        assert {a: e["effect"] for a, e in outs[0]["actions"].items()} == v["want"], (v["suite"], v["test"])
        if len(v["input"].get("actions") or []) <= 64:
            data, off = wire.pack_messages([wire.encode_check_input(v["input"])])
            for device_ingest in (True, False):
                raw, flags = ev.check_pb(data, off, device_ingest=device_ingest, **kw)
                got = wire.decode_check_output(raw[0])
                assert {a: e["effect"] for a, e in got["actions"].items()} == v["want"], (v["suite"], v["test"], device_ingest)
        compared += 1
    return compared


def test_one_image_answers_every_globals_set_on_the_simulator():
    lt = lower_rule_table(store_rule_table(), per_call_globals=True)
    assert _one_image(lambda: HostSimEvaluator(lt, Conf())) == len(VECTORS)


def test_undefined_global_is_an_error_not_a_guess():
    """`G.x` with no x in the call's globals: a CEL error (condition false), exactly as with folded globals"""
    lt = lower_rule_table(store_rule_table(), per_call_globals=True)
    ev = HostSimEvaluator(lt, Conf())
    folded = HostSimEvaluator(lower_rule_table(store_rule_table(), {}), Conf())
    for v in VECTORS[:20]:
        a = ev.check([v["input"]], now_ns=1_700_000_000_000_000_000, allow_unsupported=True, globals_={})
        b = folded.check([v["input"]], now_ns=1_700_000_000_000_000_000, allow_unsupported=True)
        assert a[0][0]["actions"] == b[0][0]["actions"]


def test_folded_table_refuses_an_override():
    ev = HostSimEvaluator(lower_rule_table(store_rule_table(), {"environment": "test"}), Conf(globals_={"environment": "test"}))
    with pytest.raises(ValueError):
        ev.check([VECTORS[0]["input"]], globals_={"environment": "prod"})


@pytest.mark.gpu
def test_one_image_answers_every_globals_set_on_the_gpu():
    lt = lower_rule_table(store_rule_table(), per_call_globals=True)
    ev = None

    def make():
        nonlocal ev
        ev = HipEvaluator(lt, Conf())
        return ev
    try:
        assert _one_image(make) == len(VECTORS)
    finally:
        if ev is not None:
            ev.close()


def test_the_three_flatteners_fill_the_globals_columns_alike():
    """Flattener(globals_=) == cbi_flatten_pb_g(globals_pb) array for array, == the device flattener value by value"""
    import wire_device_util as wu
    from cerbos_amd.flatten import Flattener
    from cerbos_amd.ingest import IngestTable
    lt = lower_rule_table(store_rule_table(), per_call_globals=True)
    inputs = [v["input"] for v in VECTORS if len(v["input"].get("actions") or []) <= 64]
    g = {"environment": "test", "nested": {"list": [1, "two", {"three": 3.5}], "flag": True}, "n": 42}
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    gpb = wire.encode_map(1, g)
    want = Flattener(lt).flatten(inputs, sort=False, globals_=g)
    it = IngestTable(lt.blob)
    have = it.flatten_pb(data, off, sort=False, globals_pb=gpb)
    it.close()
    for name in ("req_u32", "roles", "tuple_action", "col_tag", "col_val", "heap_tag", "heap_val", "str_off", "str_bytes", "str_flags"):
        assert np.array_equal(getattr(want, name), getattr(have, name)), name
    rc, wb = wu.sim_flatten(lt, data, off, globals_pb=gpb)
    assert rc == 0 and wb.stats["n_host"] == 0
    wu.assert_same_requests(lt, have, wb, True)
