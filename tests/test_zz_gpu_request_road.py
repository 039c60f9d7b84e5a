"""GPU tier (runs last: the round's last build, which this file belongs to, had no GPU minutes left - see DESIGN §4.4): serialized
CheckResourcesRequests down the device road (cbh_wire_check_requests_pb: cbh_wire_req.h's count + split kernels, the device
flattener, the decision kernels, the device assembler) against the same road fed with the CheckInputs svc.CheckResources builds
from those requests (cerbos_svc.go:274-288) - byte for byte the same CheckOutputs - and against the reference's service-level cases."""
import numpy as np
import pytest

from cerbos_amd import capi, wire, workloads
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json, store_rule_table
from test_hostsim_golden import GLOBALS

pytestmark = pytest.mark.gpu
NOW = 1_700_000_000_000_000_000


def _request_of(inputs, include_meta=False):
    return {"requestId": inputs[0].get("requestId", ""), "includeMeta": include_meta, "principal": inputs[0]["principal"],
            "resources": [{"actions": i["actions"], "resource": i["resource"]} for i in inputs]}


def _as_built_by_the_service(inputs):
    return [dict({k: v for k, v in i.items() if k != "auxData"}, principal=inputs[0]["principal"], requestId=inputs[0].get("requestId", ""))
            for i in inputs]


def test_service_level_cases_requests_in_outputs_out():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    cases = load_json("server_check_cases.json")
    table = capi.Table(lt.blob)
    try:
        reqs = [wire.encode_check_resources_request(_request_of(c["inputs"], include_meta=bool(k & 1))) for k, c in enumerate(cases)]
        outs, flags, meta = table.wire_check_requests_pb(reqs, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        assert [bool(m) for m in meta] == [bool(k & 1) for k in range(len(cases))]
        assert [len(o) for o in outs] == [len(c["inputs"]) for c in cases]
        k = 0
        for case, per_request in zip(cases, outs):
            for inp, want, raw in zip(case["inputs"], case["want"], per_request):
                assert not flags[k] & 1
                k += 1
                have = wire.decode_check_output(raw)
                assert have["resourceId"] == inp["resource"].get("id", "")
                assert {a: e["effect"] for a, e in have["actions"].items()} == want["actions"], case["name"]
                for a, m in want["meta"].items():
                    assert have["actions"][a]["policy"] == m["matchedPolicy"] and have["actions"][a]["scope"] == m["matchedScope"]
    finally:
        table.close()


@pytest.mark.parametrize("name,n", [("C2", 30_000), ("C5", 30_000)])
def test_requests_give_the_bytes_their_check_inputs_give(name, n):
    pol = getattr(workloads, name.lower() + "_policies")
    req_fn = getattr(workloads, name.lower() + "_requests")
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
    inputs = req_fn(n_requests=n).to_inputs()
    rng = np.random.default_rng(5)
    groups, k = [], 0
    while k < n:                                   # 1 .. 40 resource entries to a request, now and then a request without any
        m = int(rng.integers(0, 41)) if rng.random() < 0.9 else 0
        groups.append(inputs[k:k + m])
        k += m
    table = capi.Table(lt.blob)
    try:
        reqs = [wire.encode_check_resources_request(_request_of(g)) if g else b"" for g in groups]
        outs, flags, _ = table.wire_check_requests_pb(reqs, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        flat = [i for g in groups for i in _as_built_by_the_service(g)]
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in flat])
        want, want_flags = table.wire_check_pb(data, off, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        assert [len(o) for o in outs] == [len(g) for g in groups]
        assert [b for o in outs for b in o] == want
        assert np.array_equal(flags, want_flags)
    finally:
        table.close()


def test_a_malformed_request_is_named():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    case = load_json("server_check_cases.json")[0]
    good = wire.encode_check_resources_request(_request_of(case["inputs"]))
    table = capi.Table(lt.blob)
    try:
        with pytest.raises(capi.HipEngineError, match="malformed CheckResourcesRequest at index 2"):
            table.wire_check_requests_pb([good, good, good[:-3], good], now_ns=NOW)
        outs, _, _ = table.wire_check_requests_pb([], now_ns=NOW)
        assert outs == []
    finally:
        table.close()


def test_the_audit_trail_of_every_request_golden_store():
    """cbh_wire_check_requests_trail_pb: per request the policies its entries went through (one decision-log entry per call), against
    the oracle's union - the golden store (cbh_walk2_trail_kernel keeps its trail)."""
    from cerbos_amd.engine import Conf, HipEvaluator
    from oracle.check import EvalParams, RuleTableOracle
    rt = store_rule_table()
    ev, oracle = HipEvaluator(lower_rule_table(rt, GLOBALS), Conf(globals_=GLOBALS)), RuleTableOracle(rt)
    params = EvalParams(globals_=GLOBALS, now_ns=NOW)
    cases = load_json("server_check_cases.json") + [c for c in load_json("engine_cases.json") if not c["wantError"]]
    groups = [c["inputs"] for c in cases if not any("auxData" in i for i in c["inputs"])] + [[]]
    reqs = [wire.encode_check_resources_request(_request_of(g)) if g else b"" for g in groups]
    try:
        outs, oflags, _, trails = ev.check_requests_pb(reqs, now_ns=NOW, audit_trail=True)
        plain, plain_flags, _ = ev.check_requests_pb(reqs, now_ns=NOW)
        assert outs == plain and np.array_equal(oflags, plain_flags)
        k = compared = 0
        for g, trail in zip(groups, trails):
            flagged = any(oflags[k + j] & 1 for j in range(len(g)))
            k += len(g)
            if flagged:
                continue
            want = set()
            for i in _as_built_by_the_service(g):
                want.update(oracle.check(i, params)["effectivePolicies"])
            assert trail == sorted(want), g
            compared += 1
        assert compared > 40 and trails[-1] == []
    finally:
        ev.close()


@pytest.mark.parametrize("name,n", [("C2", 30_000), ("C5", 30_000)])
def test_the_audit_trail_of_every_request_at_size(name, n):
    """... and at size (C2: the flat trail kernels; C5: cbh_walk2_trail_kernel, the batch reordered by route on the device): the masks the
    request road returns against cbh_check_batch_trail over the host-flattened inputs with one group per request."""
    from cerbos_amd.flatten import Flattener
    pol = getattr(workloads, name.lower() + "_policies")
    req_fn = getattr(workloads, name.lower() + "_requests")
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
    inputs = req_fn(n_requests=n).to_inputs()
    rng = np.random.default_rng(11)
    groups, k = [], 0
    while k < n:
        m = int(rng.integers(1, 41))
        groups.append(inputs[k:k + m])
        k += m
    table = capi.Table(lt.blob)
    try:
        reqs = [wire.encode_check_resources_request(_request_of(g)) for g in groups]
        outs, flags, _, masks = table.wire_check_requests_pb(reqs, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES, trail=True)
        plain, _, _ = table.wire_check_requests_pb(reqs, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        assert outs == plain
        flat = [i for g in groups for i in _as_built_by_the_service(g)]
        batch = Flattener(lt).flatten(flat, "default", "")
        pre = np.arange(batch.n_requests) if batch.req_perm is None else np.asarray(batch.req_perm)
        by_input = np.repeat(np.arange(len(groups), dtype=np.uint32), [len(g) for g in groups])
        _, want = table.check_trail(batch, by_input[np.asarray(batch.vreq_input)[pre]], len(groups), now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        assert np.array_equal(masks, want) and masks.any()
    finally:
        table.close()
