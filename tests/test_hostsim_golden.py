"""CPU tier: the decision kernel's device source, compiled for the host (tests/hostsim), must
reproduce the reference's golden engine cases through the real lowering + flattening +
response assembly.  The GPU tier (test_gpu_golden.py) re-runs them on the MI355X."""
import pytest

import hostsim_api
from cerbos_amd import capi
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from helpers import load_json, norm_actions, store_rule_table

CASES = load_json("engine_cases.json")
GLOBALS = {"environment": "test"}

# Golden cases whose decision path needs CEL outside the device subset: the kernel must
# flag them (status UNSUPPORTED), never return a wrong effect silently.
EXPECT_UNSUPPORTED = set()   # (round 3: engine/case_21 - derived-role definitions comparing runtime.effectiveDerivedRoles == [] - decides on the device)


class HostSimEvaluator(HipEvaluator):
    """HipEvaluator with the table.check call routed to the host simulation."""

    def __init__(self, lt, conf):
        self.conf = conf
        self.lt = lt
        self.flattener = Flattener(lt)

        class _T:
            def check(_self, batch, now_ns=0, flags=0, want=(), device_order=False):
                return hostsim_api.check(lt, batch, now_ns, flags, device_order)

            def trace(_self, batch, now_ns=0, flags=0, capacity=None):
                return hostsim_api.trace(lt, batch, now_ns, flags, capacity)

            def check_trail(_self, batch, groups=None, n_groups=1, now_ns=0, flags=0, want=()):
                return hostsim_api.check_trail(lt, batch, groups, n_groups, now_ns, flags)

            # the device road of check_pb (cbh_wire.h) on the simulator: the GPU's flattener, decision and assembler kernels
            def wire_flatten(_self, data, offsets, default_policy_version="default", default_scope="", device_index=0, globals_pb=b""):
                import wire_device_util as wu
                rc, wb = wu.sim_flatten(lt, data, offsets, default_policy_version, default_scope, globals_pb=globals_pb)
                if rc == 1 or wb.stats["n_host"]:
                    raise capi.HostFlattenerNeeded("host flattener")
                if wb.stats["first_bad"] != 0xFFFFFFFF:
                    raise capi.HipEngineError("malformed CheckInput at index %d" % wb.stats["first_bad"])

                class _DB:
                    def close(_s):
                        pass
                db = _DB()
                db.wb, db.batch = wb, wu.to_batch(lt, wb, grouped=wb.grouped is not None)
                return db

            def launch(_self, db, now_ns=0, flags=0):
                db.res = hostsim_api.check(lt, db.batch, now_ns, flags, device_order=True)

            def wire_outputs(_self, db, cap=None):
                import wire_device_util as wu
                _self.device_road_calls = getattr(_self, "device_road_calls", 0) + 1
                return wu.sim_outputs(lt, db.res, db.wb.n, edr_is_grouped=db.wb.grouped is not None)

            # ... and of check_requests_pb (cbh_wire_req.h in front of it): split, flatten, decide, assemble - all on the simulator
            def wire_check_requests_pb(_self, requests, aux=None, now_ns=0, flags=0, default_policy_version="default", default_scope="",
                                       device_index=0, globals_pb=b"", trail=False):
                import numpy as np

                from cerbos_amd import wire
                from test_request_road import sim_split
                got = sim_split(list(requests), aux)
                if isinstance(got, int):
                    raise capi.HipEngineError("malformed CheckResourcesRequest at index %d" % got)
                msgs, first, rflags = got
                db = _self.wire_flatten(*wire.pack_messages(msgs), default_policy_version, default_scope, globals_pb=globals_pb)
                masks = None
                if trail:   # (cbh_wire_check_requests_trail_pb: one group per request, moved to where a grouped batch keeps the input)
                    by_input = np.repeat(np.arange(len(requests), dtype=np.uint32), np.diff(first).astype(np.int64))
                    groups = by_input
                    if db.wb.grouped is not None:
                        groups = np.zeros_like(by_input)
                        groups[db.wb.grouped[3]] = by_input
                    db.res, masks = hostsim_api.check_trail(lt, db.batch, groups, max(len(requests), 1), now_ns, flags)
                    masks = masks[:len(requests)]
                else:
                    _self.launch(db, now_ns, flags)
                outs, oflags = _self.wire_outputs(db)
                res = [outs[int(first[r]):int(first[r + 1])] for r in range(len(requests))], oflags, (rflags & 1).astype(bool)
                return res + (masks,) if trail else res
        self.table = _T()


@pytest.fixture(scope="module")
def evaluator():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    return HostSimEvaluator(lt, Conf(globals_=GLOBALS))


def _modes(case):
    return [False, True] if case["lenient"] is None else [case["lenient"]]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_engine_case(evaluator, case):
    for lenient in _modes(case):
        outs, bad = evaluator.check(case["inputs"], now_ns=1_700_000_000_000_000_000,
                                    lenient_scope_search=lenient, allow_unsupported=True)
        for i, (have, want) in enumerate(zip(outs, case["wantOutputs"])):
            if i in bad:
                assert case["name"] in EXPECT_UNSUPPORTED, "unexpected UNSUPPORTED in %s" % case["name"]
                continue
            assert norm_actions(have) == norm_actions(want), (case["name"], lenient)
            assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or [])


SERVER_CASES = load_json("server_check_cases.json")


@pytest.mark.parametrize("case", SERVER_CASES, ids=[c["name"] for c in SERVER_CASES])
def test_service_level_check_resources_case(evaluator, case):
    from helpers import assert_server_case
    outs, bad = evaluator.check(case["inputs"], now_ns=1_700_000_000_000_000_000, allow_unsupported=True)
    assert assert_server_case(case, outs, skip=bad) + len(bad) == len(case["inputs"])
    assert len(bad) < len(case["inputs"])


def _run_verify_vectors(make_evaluator):
    """tests/golden/verify_vectors.json (effects the reference engine returned for the policy-test-framework
    fixtures) through the product path; shared with the GPU tier."""
    import json

    from helpers import rfc3339_ns
    vectors = load_json("verify_vectors.json")
    groups = {}
    for v in vectors:
        groups.setdefault(json.dumps(v["globals"], sort_keys=True), []).append(v)
    compared = flagged = 0
    for gkey, vs in groups.items():
        ev = make_evaluator(json.loads(gkey))
        for v in vs:
            outs, bad = ev.check([v["input"]], now_ns=rfc3339_ns(v["now"]) if v["now"] else 1_700_000_000_000_000_000,
                                 lenient_scope_search=v["lenient"], strict_evaluation=v["strict"],
                                 default_policy_version=v["defaultPolicyVersion"], default_scope=v["defaultScope"],
                                 allow_unsupported=True)
            if bad:
                flagged += 1
                continue
            assert {a: e["effect"] for a, e in outs[0]["actions"].items()} == v["want"], (v["suite"], v["test"])
            compared += 1
    return compared, flagged


def test_policy_test_framework_vectors():
    def make(globals_):
        return HostSimEvaluator(lower_rule_table(store_rule_table(), globals_), Conf(globals_=globals_))
    compared, flagged = _run_verify_vectors(make)
    assert (compared, flagged) == (len(load_json("verify_vectors.json")), 0)
