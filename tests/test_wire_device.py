"""The device flattener (cerbos_amd/csrc/cbh_wire.h: serialized CheckInputs -> the batch in HBM, built by three kernels)
against the host flattener (libcerbos_ingest.so), request by request and value by value; then decisions from a
device-flattened batch against decisions from a host-flattened one.

CPU tier: the kernels' source on the host simulator (tests/hostsim).  GPU tier: the same through the C ABI
(cbh_wire_flatten, cbh_check_resident, cbh_wire_spans_download) - tests/test_gpu_wire.py."""
import numpy as np
import pytest

import hostsim_api
import wire_device_util as wu
from cerbos_amd import namer, wire, workloads
from cerbos_amd.ingest import IngestTable
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.lower.celc import LoweringError
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json, store_rule_table
from test_fuzz_parity import _policies, _requests
from test_hostsim_golden import GLOBALS


def _meta_flags(lt):
    """cbh_blob.h: the META section's flag word"""
    import struct
    blob = lt.blob
    nsec = struct.unpack_from("<I", blob, 8)[0]
    for i in range(nsec):
        sid, _cnt, off, _nb, _ = struct.unpack_from("<IIQQQ", blob, 32 + 32 * i)
        if sid == 1:
            meta = struct.unpack_from("<24I", blob, off)
            return meta[META_FLAGS_INDEX]
    raise AssertionError("no META section")


def _meta_index(name):
    import os
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "cerbos_amd", "csrc", "cbh_blob.h")).read()
    return int(re.search(name + r"\s*=\s*(\d+)", src).group(1))


META_FLAGS_INDEX = _meta_index("CBH_M_FLAGS")


def _compare(lt, inputs, default_version="default", default_scope="", **kw):
    """host flattener vs device flattener on the same messages; returns (host batch, device batch)"""
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    it = IngestTable(lt.blob)
    try:
        hb = it.flatten_pb(data, off, default_version, default_scope, sort=False)
    finally:
        it.close()
    rc, wb = wu.sim_flatten(lt, data, off, default_version, default_scope, **kw)
    assert rc == 0
    assert wb.stats["first_bad"] == 0xFFFFFFFF and wb.stats["n_host"] == 0, wb.stats
    wu.assert_same_requests(lt, hb, wb, bool(_meta_flags(lt) & wu.MF_READS_REQUEST_STRINGS))
    return hb, wb


@pytest.mark.parametrize("name", ["C2", "C3", "C5"])
def test_synthetic_workloads(name):
    pol, reqs = {"C2": (workloads.c2_policies, lambda: workloads.c2_requests(n_requests=700)),
                 "C3": (workloads.c3_policies, lambda: workloads.c3_requests(n_requests=700)),
                 "C5": (workloads.c5_policies, lambda: workloads.c5_requests(n_requests=700))}[name]
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
    hb, wb = _compare(lt, reqs().to_inputs())
    assert wb.stats["max_actions"] == int(hb.req_u32[9].max()) and wb.stats["max_roles"] == int(hb.req_u32[7].max())


def test_golden_store_inputs():
    """the reference's engine cases: principal / resource scopes, versions, nested attributes, JWT claims in auxData"""
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    inputs = [inp for case in load_json("engine_cases.json") for inp in case["inputs"]]
    assert any(root == "J" for root, _ in lt.columns)   # auxData.jwts: named tokens, viewed as name -> {"claims": {...}}
    hb, wb = _compare(lt, inputs)
    _compare(lt, inputs, default_version="v9", default_scope=".acme")
    # ... and the decision kernels (simulator) give the same answers on the two batches, derived roles included
    now = 1_700_000_000_000_000_000
    for flags in (4, 4 | 1):   # CBH_F_WANT_DERIVED_ROLES, + lenient scope search
        want = hostsim_api.check(lt, hb, now_ns=now, flags=flags, device_order=True)
        have = hostsim_api.check(lt, wu.to_batch(lt, wb), now_ns=now, flags=flags, device_order=True)
        for f in ("effect", "policy", "scope", "status", "edr"):
            assert np.array_equal(getattr(want, f), getattr(have, f)), f


def test_policy_test_framework_inputs():
    """inputs of tests/golden/verify_vectors.json: JWT claims (auxData.jwt) and named JWTs (auxData.jwts) in every path shape"""
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    inputs = [v["input"] for v in load_json("verify_vectors.json")]
    assert sum("jwts" in (i.get("auxData") or {}) for i in inputs) >= 2
    inputs.append({"principal": {"id": "p", "roles": ["employee"]}, "resource": {"kind": "leave_request", "id": "r"}, "actions": ["frobnicate"],
                   "auxData": {"jwts": {"token_a": {"claims": {"aud": "x"}}, "token_b": {}, "other": {"claims": {"customArray": ["A"]}}}}})
    inputs.append({"principal": {"id": "x", "roles": ["", "employee", ""], "scope": ".acme.hr.uk", "policyVersion": "20210210"},
                   "resource": {"kind": "leave_request", "id": "Ünïcode ✓", "scope": "acme.hr.zz.yy",
                                "attr": {"owner": None, "n": 1e308, "neg": -0.0, "deep": {"a": {"b": {"c": [1, [2, [3, {"k": "v"}]]]}}},
                                         "empty_list": [], "empty_map": {}, "": "empty key"}},
                   "actions": ["view", "", "view", "a" * 300], "auxData": {"jwt": {"iss": "cerbos", "aud": ["a", "b"], "nested": {"x": True}}}})
    inputs = [i for i in inputs if len(i.get("actions") or []) <= 64]
    _compare(lt, inputs)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_inputs(seed):
    """ragged roles / actions, nested and wrongly typed attributes, strings the table does not know"""
    rng = np.random.default_rng(10_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    inputs = [i for i in _requests(rng, 300) if len(i.get("actions") or []) <= 64]
    # a kind in the pre-0.30 form (`album:photo`) is looked up rewritten (namer.go:213-218): when the table does not hold the
    # rewritten text the device has no bytes to give it (they are not in the message) and leaves the message to the host
    for_host = [i for i in inputs if namer.sanitize(i["resource"]["kind"]) != i["resource"]["kind"]
                and namer.sanitize(i["resource"]["kind"]) not in lt.string_ids]
    if for_host:
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
        rc, wb = wu.sim_flatten(lt, data, off)
        assert rc == 0 and wb.stats["n_host"] == len(for_host) and wb.stats["first_bad"] == 0xFFFFFFFF
        inputs = [i for i in inputs if not any(i is h for h in for_host)]
    hb, wb = _compare(lt, inputs)
    # a dictionary and a heap that start far too small: the call grows them and runs the fill again - same batch
    hb2, wb2 = _compare(lt, inputs, dict_slots=16, heap=1)
    assert wb2.fill_runs > 1 or (wb2.heap_len <= 1 and wb2.dict_slots >= 16)


def test_messages_left_to_the_host_are_counted_not_guessed():
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c2_policies())))
    inputs = workloads.c2_requests(n_requests=70).to_inputs()
    inputs[3] = dict(inputs[3], actions=["a%d" % k for k in range(65)])                  # the host flattener splits it
    inputs[9] = dict(inputs[9], resource=dict(inputs[9]["resource"], kind="doc:old-form"))   # namer.go:213-218 rewrites it
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    rc, wb = wu.sim_flatten(lt, data, off)
    assert rc == 0 and wb.stats["n_host"] == 2 and wb.stats["first_bad"] == 0xFFFFFFFF
    assert wb.status[3] == 2


def test_malformed_message_is_named():
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c2_policies())))
    msgs = [wire.encode_check_input(i) for i in workloads.c2_requests(n_requests=130).to_inputs()]
    msgs[70] = msgs[70][:-3]          # a length that runs past the end
    msgs[101] = b"\x0f" + msgs[101]   # wire type 7
    data, off = wire.pack_messages(msgs)
    rc, wb = wu.sim_flatten(lt, data, off)
    assert rc == 0 and wb.stats["first_bad"] == 70


def test_empty_batch_and_empty_messages():
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c2_policies())))
    rc, wb = wu.sim_flatten(lt, np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert rc == 0 and wb.n == 0 and wb.n_tuples == 0
    _compare(lt, [{}, {"actions": ["view"]}, {"principal": {"id": "x", "roles": []}, "resource": {"kind": ""}, "actions": ["", "view", ""]}])


@pytest.mark.parametrize("name", ["C2", "C5"])
def test_decisions_from_a_device_flattened_batch(name):
    """the decision kernels (simulator) on the device-flattened batch == on the host-flattened batch, tuple by tuple"""
    pol, reqs = {"C2": (workloads.c2_policies, lambda: workloads.c2_requests(n_requests=400)),
                 "C5": (workloads.c5_policies, lambda: workloads.c5_requests(n_requests=400))}[name]
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
    hb, wb = _compare(lt, reqs().to_inputs())
    now = 1_700_000_000_000_000_000
    want = hostsim_api.check(lt, hb, now_ns=now, device_order=True)
    have = hostsim_api.check(lt, wu.to_batch(lt, wb), now_ns=now, device_order=True)
    for f in ("effect", "policy", "scope", "status"):
        assert np.array_equal(getattr(want, f), getattr(have, f)), f


def test_outputs_assembled_from_the_device_spans():
    """cbi_assemble_wire_pb (spans the fill kernel noted) == cbi_assemble_pb (spans the host flattener noted), byte for byte"""
    from cerbos_amd import capi
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
    inputs = workloads.c5_requests(n_requests=300).to_inputs()
    inputs[5] = dict(inputs[5], actions=["view", "view", "edit"], requestId="")     # duplicate actions, no request id
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    it = IngestTable(lt.blob)
    hb = it.flatten_pb(data, off, sort=False)
    rc, wb = wu.sim_flatten(lt, data, off)
    assert rc == 0
    rng = np.random.default_rng(5)
    res = capi.Result(hb.n_tuples, hb.n_requests, ("policy", "scope", "status", "edr"))
    res.effect[:] = rng.integers(1, 3, hb.n_tuples)
    res.policy[:] = (2 << 28)   # resource policy at the root scope
    res.scope[:] = 0xFFFFFFFF
    res.status[:] = rng.integers(0, 2, hb.n_tuples)
    res.edr[:] = rng.integers(0, 4, hb.n_requests)
    want, wflags = it.assemble_pb(hb, res, data, off)
    act_off = np.concatenate([wb.req_u32[8], [wb.n_tuples]]).astype(np.uint32)
    have, hflags = it.assemble_wire_pb(res, data, off, (np.ascontiguousarray(wb.in_span), np.ascontiguousarray(wb.act_span), act_off))
    assert have == want and np.array_equal(wflags, hflags)
    it.close()


def _random_results(rng, lt, n_tuples, n_requests):
    from cerbos_amd import capi
    res = capi.Result(n_tuples, n_requests, ("policy", "scope", "status", "edr"))
    res.effect[:] = rng.integers(0, 3, n_tuples)
    kinds = rng.integers(0, 6, n_tuples)   # enum cbh_policy_kind
    ident = np.where(kinds == 4, rng.integers(0, max(1, len(lt.policy_keys)), n_tuples), rng.integers(0, max(1, len(lt.scopes)), n_tuples))
    if not lt.policy_keys:
        kinds[kinds == 4] = 1
    res.policy[:] = (kinds.astype(np.uint32) << 28) | ident.astype(np.uint32)
    res.scope[:] = np.where(rng.random(n_tuples) < 0.5, 0xFFFFFFFF, rng.integers(0, max(1, len(lt.scopes)), n_tuples)).astype(np.uint32)
    res.status[:] = rng.integers(0, 4, n_tuples)
    res.edr[:] = rng.integers(0, 1 << min(63, max(1, len(lt.dr_names))), n_requests).astype(np.uint64)
    return res


@pytest.mark.parametrize("name", ["C3", "C5", "fuzz0", "fuzz3"])
def test_outputs_written_by_the_device(name):
    """cbh_wire_out_* (the GPU writes the CheckOutputs) == cbi_assemble_wire_pb (a host thread does), byte for byte, on results
    that take every branch: all policy word kinds, scopes or none, every status, duplicate actions, old-form kinds and versions"""
    rng = np.random.default_rng(77)
    if name.startswith("fuzz"):
        frng = np.random.default_rng(10_000 + int(name[4:]))
        try:
            lt = lower_rule_table(rule_table_from_policies(policies_from_docs(_policies(frng))))
        except LoweringError:
            pytest.skip("store refused by the lowering")
        inputs = [i for i in _requests(frng, 300) if len(i.get("actions") or []) <= 64]
    else:
        pol, reqs = {"C3": (workloads.c3_policies, workloads.c3_requests), "C5": (workloads.c5_policies, workloads.c5_requests)}[name]
        lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
        inputs = reqs(n_requests=300).to_inputs()
    inputs[5] = dict(inputs[5], actions=["view", "view", "edit", "view"], requestId="")
    inputs[6] = dict(inputs[6], resource=dict(inputs[6]["resource"], policyVersion="2024-01:beta"), principal=dict(inputs[6]["principal"], policyVersion="x y"))
    inputs[7] = dict(inputs[7], actions=[])
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    rc, wb = wu.sim_flatten(lt, data, off, default_version="v:old-form")
    assert rc == 0
    it = IngestTable(lt.blob)
    res = _random_results(rng, lt, wb.n_tuples, wb.n)
    act_off = np.concatenate([wb.req_u32[8], [wb.n_tuples]]).astype(np.uint32)
    want, wflags = it.assemble_wire_pb(res, data, off, (np.ascontiguousarray(wb.in_span), np.ascontiguousarray(wb.act_span), act_off), "v:old-form")
    have, hflags = wu.sim_outputs(lt, res, wb.n)
    for i, (a, b) in enumerate(zip(want, have)):
        assert a == b, (i, wire.decode_check_output(a), wire.decode_check_output(b))
    assert np.array_equal(wflags, hflags)
    it.close()


def test_mutated_messages_both_flatteners_agree():
    """Random corruptions of valid messages (bit flips, cuts, splices, inserted bytes): the device flattener and the host
    flattener must agree on WHICH are malformed, and on the batch when they are not - and neither may crash."""
    from cerbos_amd.ingest import IngestError
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    inputs = [inp for case in load_json("engine_cases.json") for inp in case["inputs"] if len(inp.get("actions") or []) <= 64][:40]
    msgs = [wire.encode_check_input(i) for i in inputs]
    rng = np.random.default_rng(11)
    ok = bad = 0
    for trial in range(700):
        m = bytearray(msgs[int(rng.integers(0, len(msgs)))])
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 4))
            pos = int(rng.integers(0, len(m))) if m else 0
            if kind == 0 and m:
                m[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                del m[pos:pos + int(rng.integers(1, 9))]
            elif kind == 2:
                m[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 6)), dtype=np.uint8))
            else:
                other = msgs[int(rng.integers(0, len(msgs)))]
                m[pos:pos + 8] = other[:8]
        neighbours = [msgs[0], bytes(m), msgs[1]]
        data, off = wire.pack_messages(neighbours)
        try:
            hb = it.flatten_pb(data, off, sort=False)
            host_bad = False
        except IngestError:
            host_bad = True
        rc, wb = wu.sim_flatten(lt, data, off)
        assert rc == 0
        dev_bad = wb.stats["first_bad"] != 0xFFFFFFFF
        assert dev_bad == host_bad, (trial, bytes(m))
        if host_bad:
            assert wb.stats["first_bad"] == 1
            bad += 1
        elif wb.stats["n_host"] == 0 and hb.n_requests == 3:
            wu.assert_same_requests(lt, hb, wb, bool(_meta_flags(lt) & wu.MF_READS_REQUEST_STRINGS))
            ok += 1
    it.close()
    assert ok > 60 and bad > 300, (ok, bad)


def test_nesting_and_width_boundaries():
    """containers nested 8 deep flatten on the device, 9 deep are the host's; wide lists / maps; empty containers; a key twice"""
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
    root_cols = [keys for root, keys in lt.columns if root == "R"]
    assert root_cols, "C5 reads resource attributes"
    name = root_cols[0][0]

    def nest(d):
        v = "leaf"
        for k in range(d):   # one level per step: list and map in turn, with siblings
            v = [k, v, None] if k % 2 else {"a": v, "b": True}
        return v
    base = workloads.c5_requests(n_requests=4).to_inputs()
    ok = []
    for d in (1, 3, 8):
        i = dict(base[0], resource=dict(base[0]["resource"], attr=dict(base[0]["resource"].get("attr") or {}, **{name: nest(d)})))
        ok.append(i)
    ok.append(dict(base[1], resource=dict(base[1]["resource"], attr={name: list(range(300)), "other": {"k%d" % k: [k] for k in range(120)}})))
    ok.append(dict(base[2], resource=dict(base[2]["resource"], attr={name: [], "m": {}, "": {"": [[]]}})))
    hb, wb = _compare(lt, ok)
    assert wb.heap_len > 300
    deep = dict(base[3], resource=dict(base[3]["resource"], attr={name: nest(9)}))
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in ok + [deep]])
    rc, wb2 = wu.sim_flatten(lt, data, off)
    assert rc == 0 and wb2.stats["n_host"] == 1 and wb2.stats["first_bad"] == 0xFFFFFFFF


@pytest.mark.parametrize("name", ["C3", "C5", "fuzz1"])
def test_grouping_by_route_on_the_device(name, monkeypatch):
    """cbh_wire_route / route_scan / gather kernels: a permutation that puts equal routes side by side; the decision kernels give
    the same answers on the grouped batch (tuples never move; derived-role masks follow their request), the assembler the same bytes"""
    monkeypatch.setenv("CBH_WIRE_GROUP", "1")
    if name.startswith("fuzz"):
        frng = np.random.default_rng(10_000 + int(name[4:]))
        try:
            lt = lower_rule_table(rule_table_from_policies(policies_from_docs(_policies(frng))))
        except LoweringError:
            pytest.skip("store refused by the lowering")
        inputs = [i for i in _requests(frng, 500) if len(i.get("actions") or []) <= 64]
    else:
        pol, reqs = {"C3": (workloads.c3_policies, workloads.c3_requests), "C5": (workloads.c5_policies, workloads.c5_requests)}[name]
        lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
        inputs = reqs(n_requests=900).to_inputs()
    rng = np.random.default_rng(3)
    inputs = [inputs[k] for k in rng.permutation(len(inputs))]   # arrival order: routes thoroughly mixed
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    rc, wb = wu.sim_flatten(lt, data, off)
    assert rc == 0 and wb.grouped is not None
    req_g, tag_g, val_g, inv, n_routes = wb.grouped
    n = wb.n
    assert sorted(inv.tolist()) == list(range(n))
    assert np.array_equal(req_g[:, inv], wb.req_u32) and np.array_equal(tag_g[:, inv], wb.col_tag) and np.array_equal(val_g[:, inv], wb.col_val)
    def route(r, req):
        return (int(req[3, r]), int(req[5, r]), int(req[4, r]))   # kind, version, scope
    routes_in_order = [route(r, req_g) for r in range(n)]
    distinct = len(set(routes_in_order))
    changes = 1 + sum(routes_in_order[r] != routes_in_order[r - 1] for r in range(1, n))
    assert n_routes == distinct == changes, (n_routes, distinct, changes)   # every route one contiguous run
    assert distinct >= 3
    now = 1_700_000_000_000_000_000
    plain = hostsim_api.check(lt, wu.to_batch(lt, wb), now_ns=now, flags=4, device_order=True)
    grouped = hostsim_api.check(lt, wu.to_batch(lt, wb, grouped=True), now_ns=now, flags=4, device_order=True)
    for f in ("effect", "policy", "scope", "status"):
        assert np.array_equal(getattr(plain, f), getattr(grouped, f)), f
    assert np.array_equal(grouped.edr[inv], plain.edr)
    a, af = wu.sim_outputs(lt, plain, n)
    b, bf = wu.sim_outputs(lt, grouped, n, edr_is_grouped=True)
    assert a == b and np.array_equal(af, bf)


def test_bytes_road_on_the_simulator_with_grouping(monkeypatch):
    """HipEvaluator.check_pb on the simulator: device road (flattener, routing, decision, assembler kernels) == host road on a
    stream of mixed routes large enough to be grouped"""
    from cerbos_amd.engine import Conf
    from test_hostsim_golden import HostSimEvaluator
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
    inputs = workloads.c5_requests(n_requests=400).to_inputs()
    rng = np.random.default_rng(9)
    inputs = [inputs[k] for k in rng.permutation(len(inputs))]
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    ev = HostSimEvaluator(lt, Conf())
    a = ev.check_pb(data, off, now_ns=1_700_000_000_000_000_000)
    assert ev.last_road == "device"
    b = ev.check_pb(data, off, now_ns=1_700_000_000_000_000_000, device_ingest=False)
    assert ev.last_road == "host" and a[0] == b[0] and list(a[1]) == list(b[1])


def test_the_whole_token_as_a_value():
    """Conditions that take the claims of the verified token - or the map of named tokens - as ONE value (membership, size, indexing
    by a computed key) make the column the whole map: the device flattener builds it on the heap (the claims under "claims", as the
    reference's `request.aux_data.jwt` view has them; named tokens as name -> {"claims": ..}) - shapes the round's coverage table showed
    no test reached.  Host flattener against device flattener value by value, then the decisions on both batches against the oracle."""
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies
    from cerbos_amd import capi
    from oracle.check import EvalParams, RuleTableOracle
    conds = {"has_claim": '"aud" in request.aux_data.jwt', "n_claims": "size(request.auxData.jwt) == 3",
             "by_key": 'request.aux_data.jwt[R.attr.claim] == "cerbos"', "named": '"partner" in request.auxData.jwts',
             "named_claim": 'request.auxData.jwts[R.attr.token].claims.iss == "acme"'}
    docs = [{"apiVersion": "api.cerbos.dev/v1", "resourcePolicy": {"resource": "tok", "version": "default", "rules": [
        {"actions": [n], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": e}}} for n, e in conds.items()]}}]
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    auxes = [{"jwt": {"aud": "cerbos", "iss": "acme", "n": 3}, "jwts": {"partner": {"iss": "acme", "k": [1, 2]}, "other": {"iss": "x"}}},
             {"jwt": {"iss": "acme"}, "jwts": {"other": {"iss": "acme"}}},
             {"jwt": {"aud": ["a", "b"], "sub": "s", "nested": {"a": {"b": 1}}}},
             {"jwts": {"partner": {}}},
             None]
    inputs = []
    for k, aux in enumerate(auxes):
        for claim, token in (("aud", "partner"), ("iss", "other"), ("none", "none")):
            inp = {"requestId": "t%d" % k, "actions": list(conds), "resource": {"kind": "tok", "id": "r", "attr": {"claim": claim, "token": token}},
                   "principal": {"id": "p", "roles": ["user"], "attr": {}}}
            if aux is not None:
                inp["auxData"] = aux
            inputs.append(inp)
    hb, wb = _compare(lt, inputs)
    now = 1_700_000_000_000_000_000
    want = hostsim_api.check(lt, hb, now_ns=now, flags=4, device_order=True)
    have = hostsim_api.check(lt, wu.to_batch(lt, wb), now_ns=now, flags=4, device_order=True)
    for f in ("effect", "policy", "scope", "status", "edr"):
        assert np.array_equal(getattr(want, f), getattr(have, f)), f
    res = hostsim_api.check(lt, hb, now_ns=now, flags=4)
    orc, at, decided = RuleTableOracle(rt), 0, 0
    for inp in inputs:
        out = orc.check(inp, EvalParams(now_ns=now))
        for a in inp["actions"]:
            if res.status[at] != capi.ST_UNSUPPORTED:
                assert (res.effect[at] == 1) == (out["actions"][a]["effect"] == "EFFECT_ALLOW"), (a, inp.get("auxData"), inp["resource"]["attr"])
                decided += 1
            at += 1
    assert decided > 40, (decided, lt.unsupported)


def _raw_fields(buf):
    """(field number, wire type, value or payload bytes) of a message; raises ValueError on malformed input"""
    i, n = 0, len(buf)

    def varint():
        nonlocal i
        r = sh = 0
        while True:
            if i >= n or sh >= 70:
                raise ValueError("varint")
            b = buf[i]
            i += 1
            r |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                return r & ((1 << 64) - 1)
    while i < n:
        key = varint()
        num, wt = key >> 3, key & 7
        if wt == 0:
            yield num, wt, varint()
        elif wt == 2:
            ln = varint()
            if ln > n - i:
                raise ValueError("length")
            yield num, wt, bytes(buf[i:i + ln])
            i += ln
        elif wt == 1:
            yield num, wt, bytes(buf[i:i + 8]); i += 8
        elif wt == 5:
            yield num, wt, bytes(buf[i:i + 4]); i += 4
        else:
            raise ValueError("wire type")


# which length-delimited fields of which message are messages themselves (engine.proto CheckInput and google.protobuf.Value)
_SUB = {"CheckInput": {2: "Party", 3: "Party", 5: "AuxData"}, "Party": {4: "Entry"}, "Entry": {2: "Value"}, "Value": {5: "Struct", 6: "List"},
        "Struct": {1: "Entry"}, "List": {1: "Value"}, "AuxData": {1: "Entry", 2: "JwtsEntry"}, "JwtsEntry": {2: "JWT"}, "JWT": {1: "Entry"}}


def _overlong(buf, rng, ctx="CheckInput"):
    """The same message with every tag, length and varint value - its sub-messages' too - written as a LONGER varint than needed
    (padding continuation bytes: legal protobuf, nothing a marshaller emits), up to ten bytes."""
    def varint(v, total):
        return bytes(((v >> (7 * k)) & 0x7F) | (0x80 if k + 1 < total else 0) for k in range(total))

    def padded(v):
        need = max(1, (v.bit_length() + 6) // 7)
        return varint(v, int(rng.integers(need, 11)))      # its own length .. ten bytes

    out = bytearray()
    for num, wt, v in _raw_fields(buf):
        out += padded(num << 3 | wt)
        if wt == 0:
            out += padded(v)
        elif wt == 2:
            sub = _SUB.get(ctx, {}).get(num)
            body = _overlong(v, rng, sub) if sub else v
            out += padded(len(body)) + body
        else:
            out += v
    return bytes(out)


def test_varints_longer_than_they_need_to_be():
    """Tags, lengths and values padded to up to ten bytes: the device flattener (an eight-byte window per round trip, the ninth and tenth
    byte fetched apart) reads them as the host flattener (byte by byte) does; an eleventh byte, or a varint cut off by its message's end,
    is malformed for both."""
    from cerbos_amd.ingest import IngestError
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    inputs = [inp for case in load_json("engine_cases.json") for inp in case["inputs"] if len(inp.get("actions") or []) <= 64][:30]
    msgs = [wire.encode_check_input(i) for i in inputs]
    rng = np.random.default_rng(5)
    same = refused = 0
    for trial in range(160):
        plain = msgs[trial % len(msgs)]
        m = _overlong(plain, rng)
        assert [(a, b) for a, b, _ in _raw_fields(m)] == [(a, b) for a, b, _ in _raw_fields(plain)] and len(m) > len(plain)
        kind = trial % 4
        if kind == 1:      # an eleven-byte tag in front: malformed
            m = b"\x80" * 10 + b"\x01" + m
        elif kind == 2:    # the last varint cut off by the message's end
            m = m + b"\x88\x80"
        data, off = wire.pack_messages([msgs[0], m, msgs[1]])
        try:
            hb = it.flatten_pb(data, off, sort=False)
            host_bad = False
        except IngestError:
            host_bad = True
        rc, wb = wu.sim_flatten(lt, data, off)
        assert rc == 0
        dev_bad = wb.stats["first_bad"] != 0xFFFFFFFF
        assert dev_bad == host_bad == (kind in (1, 2)), (trial, kind, m)
        if host_bad:
            assert wb.stats["first_bad"] == 1
            refused += 1
        else:
            data0, off0 = wire.pack_messages([msgs[0], plain, msgs[1]])
            rc0, wb0 = wu.sim_flatten(lt, data0, off0)
            assert rc0 == 0 and wb0.stats["n_host"] == wb.stats["n_host"]
            if wb.stats["n_host"] == 0 and hb.n_requests == 3:
                wu.assert_same_requests(lt, hb, wb, bool(_meta_flags(lt) & wu.MF_READS_REQUEST_STRINGS))
                same += 1
    it.close()
    assert same > 40 and refused > 60, (same, refused)
