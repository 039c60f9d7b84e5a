"""CPU tier: the C++ wire-format ingest (libcerbos_ingest.so) against the Python
flattener that defines the ``cbh_batch`` contract - every array of the batch must be identical, with and
without the routing sort - and end to end: reference golden cases through wire bytes -> C++ ingest -> the
decision kernel's source on the host simulator."""
import numpy as np
import pytest

from cerbos_amd import wire, workloads
from cerbos_amd.engine import Conf
from cerbos_amd.flatten import Flattener
from cerbos_amd.ingest import IngestError, IngestTable, WireFlattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.lower.celc import LoweringError
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json, norm_actions, store_rule_table
from test_fuzz_parity import _policies, _requests
from test_hostsim_golden import GLOBALS, HostSimEvaluator

ARRAYS = ("req_u32", "roles", "tuple_req", "tuple_action", "col_tag", "col_val", "heap_tag", "heap_val",
          "str_off", "str_bytes", "str_flags")


def _same(lt, inputs, **kw):
    it = IngestTable(lt.blob)
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    for sort in (False, True):
        want = Flattener(lt).flatten(inputs, sort=sort, **kw)
        have = it.flatten_pb(data, off, kw.get("default_policy_version", "default"), kw.get("default_scope", ""), sort)
        assert (have.n_requests, have.n_tuples, have.n_strings) == (want.n_requests, want.n_tuples, want.n_strings)
        for name in ARRAYS:
            a, b = getattr(have, name), getattr(want, name)
            assert a.dtype == b.dtype and a.shape == b.shape, (name, sort, a.shape, b.shape)
            assert np.array_equal(a, b), (name, sort)
        tp = want.tuple_perm if want.tuple_perm is not None else np.arange(want.n_tuples)
        assert np.array_equal(have.tuple_perm, tp), sort
        if len(inputs) >= 2048:     # the in-call parallel flatten must give the very same batch
            for threads in (2, 5):
                par = it.flatten_pb(data, off, kw.get("default_policy_version", "default"), kw.get("default_scope", ""), sort, threads)
                for name in ARRAYS + ("tuple_perm", "vreq_input"):
                    assert np.array_equal(getattr(par, name), getattr(have, name)), (name, sort, threads)
        vi = want.vreq_input if want.req_perm is None else want.vreq_input[want.req_perm]
        assert np.array_equal(have.vreq_input, vi), sort
    it.close()


def test_golden_store_inputs_flatten_identically():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    inputs = [inp for case in load_json("engine_cases.json") for inp in case["inputs"]]
    assert len(inputs) > 50
    _same(lt, inputs)
    _same(lt, inputs, default_policy_version="v9", default_scope=".acme")


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_inputs_flatten_identically(seed):
    rng = np.random.default_rng(10_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    _same(lt, _requests(rng, 300))     # ragged roles / actions, > 64 actions, nested and wrongly typed attributes
    if seed < 3:
        _same(lt, _requests(rng, 5200))    # large enough for the in-call parallel flatten (slices of >= 1024)


@pytest.mark.parametrize("name", ["C2", "C3", "C5"])
def test_synthetic_workloads_flatten_identically(name):
    pol, reqs = {"C2": (workloads.c2_policies, lambda: workloads.c2_requests(n_requests=5000)),
                 "C3": (workloads.c3_policies, lambda: workloads.c3_requests(n_requests=5000)),
                 "C5": (workloads.c5_policies, lambda: workloads.c5_requests(n_requests=5000))}[name]
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
    _same(lt, reqs().to_inputs())


def test_policy_test_framework_inputs_flatten_identically():
    """Inputs of tests/golden/verify_vectors.json: JWT claims (auxData.jwt) and named JWTs (auxData.jwts)."""
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    assert any(root == "S" for root, _ in lt.columns) and any(root == "J" for root, _ in lt.columns)
    inputs = [v["input"] for v in load_json("verify_vectors.json")]
    assert sum("jwts" in (i.get("auxData") or {}) for i in inputs) >= 2
    inputs.append({"principal": {"id": "p", "roles": ["employee"]}, "resource": {"kind": "leave_request", "id": "r"}, "actions": ["frobnicate"],
                   "auxData": {"jwts": {"token_a": {"claims": {"aud": "x"}}, "token_b": {}, "other": {"claims": {"customArray": ["A"]}}}}})
    _same(lt, inputs)


def test_edge_cases():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    weird = [
        {"principal": {"id": "", "roles": []}, "resource": {"kind": "", "id": ""}, "actions": []},
        {"principal": {"id": "x", "roles": ["", "employee", ""], "scope": ".acme.hr.uk", "policyVersion": "20210210"},
         "resource": {"kind": "leave_request:special-kind/x", "id": "Ünïcode ✓", "scope": "acme.hr.zz.yy",
                      "attr": {"owner": None, "n": 1e308, "neg": -0.0, "deep": {"a": {"b": {"c": [1, [2, [3, {"k": "v"}]]]}}},
                               "empty_list": [], "empty_map": {}, "": "empty key"}},
         "actions": ["view", "", "view", "a" * 300], "auxData": {"jwt": {"iss": "cerbos", "aud": ["a", "b"], "nested": {"x": True}}}},
        {"principal": {"id": "y", "roles": ["r%d" % i for i in range(200)]}, "resource": {"kind": "leave_request", "id": "1"},
         "actions": ["act%d" % i for i in range(129)]},
    ]
    # namer.SanitizedResource: only names of the pre-0.30 form are rewritten (namer.go:213-218)
    weird += [{"principal": {"id": "z", "roles": ["employee"]}, "resource": {"kind": k, "id": "1"}, "actions": ["view"]}
              for k in ("a-b\n", "9abc-x", "a::b", "a:b:", "a@b/c-d:e.f", "x--y@@z", "ü-x", ":a")]
    _same(lt, weird)
    # old-form kinds that share length and the first eight bytes of their rewritten form: the rewritten text lives in a
    # scratch string the ingest reuses - it must not be answered from the cache in front of the string dictionaries
    alike = [{"principal": {"id": "z", "roles": ["employee"]}, "resource": {"kind": k, "id": "1"}, "actions": ["view"]}
             for k in ("doc:public_aa", "doc:public_ab", "doc:public_aa", "doc-public@ac", "doc/public-ad", "doc:public_ab")]
    _same(lt, alike)
    _same(lt, [])


def test_malformed_input_is_rejected():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    good = wire.encode_check_input({"principal": {"id": "p", "roles": ["a"], "attr": {"k": [1, 2, {"z": "w"}]}},
                                    "resource": {"kind": "leave_request", "id": "r", "attr": {"owner": "p"}}, "actions": ["view"]})
    for cut in range(1, len(good)):      # every truncation either parses as a shorter message or is refused, never crashes
        data, off = wire.pack_messages([good[:cut]])
        try:
            it.flatten_pb(data, off)
        except IngestError:
            pass
    with pytest.raises(IngestError):
        it.flatten_pb(*wire.pack_messages([b"\x12\xff\xff\xff\xff\x0f"]))   # length runs past the end
    with pytest.raises(IngestError):
        IngestTable(lt.blob[:100])
    with pytest.raises(IngestError):
        IngestTable(b"\0" * 4096)


def test_golden_cases_through_the_wire_format():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    ev = HostSimEvaluator(lt, Conf(globals_=GLOBALS))
    ev.flattener = WireFlattener(lt)
    compared = 0
    for case in load_json("engine_cases.json"):
        for lenient in ([False, True] if case["lenient"] is None else [case["lenient"]]):
            outs, bad = ev.check(case["inputs"], now_ns=1_700_000_000_000_000_000, lenient_scope_search=lenient, allow_unsupported=True)
            for i, (have, want) in enumerate(zip(outs, case["wantOutputs"])):
                if i in bad:
                    continue
                assert norm_actions(have) == norm_actions(want), (case["name"], lenient)
                assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or [])
                compared += 1
    assert compared > 60


# ---- response assembly: device results -> serialized CheckOutput, against engine.assemble ------
def _assemble_both(lt, inputs, conf, flags_extra=0, dver="default"):
    import copy

    import hostsim_api
    from cerbos_amd import capi
    it = IngestTable(lt.blob)
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    batch = it.flatten_pb(data, off, dver, "")
    batch.actions_per_request = [list(inp.get("actions") or []) for inp in inputs]
    res = hostsim_api.check(lt, batch, 1_700_000_000_000_000_000, capi.F_WANT_DERIVED_ROLES | flags_extra, device_order=True)
    raw, flags = it.assemble_pb(batch, res, data, off, dver)
    if len(inputs) >= 2048:     # assembled on several threads: the very same bytes
        raw4, flags4 = it.assemble_pb(batch, res, data, off, dver, threads=4)
        assert raw4 == raw and np.array_equal(flags4, flags)
    have = [wire.decode_check_output(r) for r in raw]
    ev = HostSimEvaluator(lt, conf)
    want, bad = ev.assemble(inputs, batch, copy.copy(res).to_input_order(batch), dver, allow_unsupported=True)
    assert sorted(bad) == [i for i, f in enumerate(flags) if f & 1]
    return have, want, flags


def test_assembly_matches_python_on_golden_cases():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    inputs = [inp for case in load_json("engine_cases.json") for inp in case["inputs"]]
    for dver in ("default", "20210210"):
        have, want, flags = _assemble_both(lt, inputs, Conf(globals_=GLOBALS), dver=dver)
        assert have == want
    assert any(f & 2 for f in flags)      # the golden store has cases with absorbed CEL errors


@pytest.mark.parametrize("seed", range(6))
def test_assembly_matches_python_on_fuzzed_stores(seed):
    from cerbos_amd import capi
    rng = np.random.default_rng(10_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    inputs = _requests(rng, 200 if seed else 4500)
    inputs[3]["actions"] = ["view", "edit", "view", "view", "edit"]      # duplicates fold DENY-sticky
    inputs[4]["actions"] = []
    for extra in (0, capi.F_LENIENT_SCOPE_SEARCH):
        have, want, _ = _assemble_both(lt, inputs, Conf(), extra)
        assert have == want


def test_assembly_rejects_a_foreign_batch():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    inputs = [inp for case in load_json("engine_cases.json")[:3] for inp in case["inputs"]]
    import hostsim_api
    it = IngestTable(lt.blob)
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    batch = it.flatten_pb(data, off)
    res = hostsim_api.check(lt, batch, 0, 0, device_order=True)
    d2, o2 = wire.pack_messages([wire.encode_check_input(i) for i in inputs[:-1]])
    with pytest.raises(IngestError):
        it.assemble_pb(batch, res, d2, o2)


def test_bytes_in_bytes_out_on_golden_cases():
    """HipEvaluator.check_pb (wire -> ingest -> kernel -> assembly) against the reference's golden outputs."""
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    ev = HostSimEvaluator(lt, Conf(globals_=GLOBALS))
    compared = 0
    for case in load_json("engine_cases.json"):
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in case["inputs"]])
        for lenient in ([False, True] if case["lenient"] is None else [case["lenient"]]):
            raw, flags = ev.check_pb(data, off, now_ns=1_700_000_000_000_000_000, lenient_scope_search=lenient)
            for have, want, f, inp in zip(raw, case["wantOutputs"], flags, case["inputs"]):
                if f & 1:
                    continue
                have = wire.decode_check_output(have)
                assert norm_actions(have) == norm_actions(want), (case["name"], lenient)
                assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or [])
                assert have["requestId"] == inp.get("requestId", "") and have["resourceId"] == inp["resource"].get("id", "")
                compared += 1
            assert ev.last_road == "device"   # the GPU's flattener / assembler kernels (simulated) produced these bytes
    assert compared > 60
    # ... and the host road gives the same bytes
    for case in load_json("engine_cases.json")[:12]:
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in case["inputs"]])
        a = ev.check_pb(data, off, now_ns=1_700_000_000_000_000_000, trace=True)
        b = ev.check_pb(data, off, now_ns=1_700_000_000_000_000_000, trace=True, device_ingest=False)
        assert ev.last_road == "host" and a[0] == b[0] and list(a[1]) == list(b[1]), case["name"]


def test_mutated_messages_never_crash():
    """The ingest takes bytes from the network side: random corruptions of valid messages (bit flips, cuts,
    splices, inserted bytes) must end in an error or in some batch - never in a crash or a hang.  (Run under
    ASan/UBSan when changing the parser: see the build line in DESIGN.md.)"""
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    inputs = [inp for case in load_json("engine_cases.json") for inp in case["inputs"]][:40]
    msgs = [wire.encode_check_input(i) for i in inputs]
    rng = np.random.default_rng(7)
    ok = bad = 0
    for trial in range(1500):
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
                cut = int(rng.integers(0, len(other)))
                m[pos:] = other[cut:]
        batch = [bytes(m), msgs[trial % len(msgs)]]
        data, off = wire.pack_messages(batch)
        try:
            b = it.flatten_pb(data, off)
            assert b.n_requests >= 2 and b.tuple_action.size == b.n_tuples
            ok += 1
        except IngestError:
            bad += 1
    assert ok > 100 and bad > 100, (ok, bad)


# ---- one serialized CheckResourcesRequest in, one serialized CheckResourcesResponse out --------------------------
def _request_of(inputs, include_meta=True):
    return {"requestId": inputs[0].get("requestId", ""), "includeMeta": include_meta, "principal": inputs[0]["principal"],
            "resources": [{"actions": i["actions"], "resource": i["resource"]} for i in inputs]}


def _same_as_check_inputs(lt, inputs, aux=None, threads=1):
    """cbi_flatten_request_pb must give the very batch that the equivalent CheckInputs give."""
    it = IngestTable(lt.blob)
    ins = [dict(i, principal=inputs[0]["principal"], requestId=inputs[0].get("requestId", ""), **({"auxData": aux} if aux else {}))
           for i in inputs]
    for i in ins:
        if not aux:
            i.pop("auxData", None)
    want = it.flatten_pb(*wire.pack_messages([wire.encode_check_input(i) for i in ins]))
    have = it.flatten_request_pb(wire.encode_check_resources_request(_request_of(ins)), wire.encode_aux_data(aux) if aux else None,
                                 threads=threads)
    for name in ARRAYS + ("tuple_perm", "vreq_input"):
        assert np.array_equal(getattr(have, name), getattr(want, name)), name
    return it, ins, have


def test_request_form_flattens_like_its_check_inputs():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    for case in load_json("server_check_cases.json"):
        _same_as_check_inputs(lt, case["inputs"])
    jwt_inputs = [v["input"] for v in load_json("verify_vectors.json") if "auxData" in v["input"]]
    _same_as_check_inputs(lt, jwt_inputs[:1], aux=jwt_inputs[0]["auxData"])
    rng = np.random.default_rng(3)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    lt2 = lower_rule_table(rt)
    _same_as_check_inputs(lt2, _requests(rng, 4300), threads=4)      # many resource entries: the in-call parallel path


@pytest.mark.parametrize("case", load_json("server_check_cases.json"), ids=lambda c: c["name"])
def test_service_level_cases_request_in_response_out(case):
    """The reference's CheckResources cases as the service sees them: CheckResourcesRequest bytes in,
    CheckResourcesResponse bytes out (libcerbos_ingest.so around the kernel source)."""
    import hostsim_api
    from cerbos_amd import capi
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    for include_meta in (True, False):
        req = wire.encode_check_resources_request(_request_of(case["inputs"], include_meta))
        batch = it.flatten_request_pb(req)
        res = hostsim_api.check(lt, batch, 1_700_000_000_000_000_000, capi.F_WANT_DERIVED_ROLES, device_order=True)
        raw, flags = it.assemble_response_pb(batch, res, req)
        resp = wire.decode_check_resources_response(raw)
        assert resp["requestId"] == case["inputs"][0].get("requestId", "") and len(resp["results"]) == len(case["inputs"])
        for inp, have, want, f in zip(case["inputs"], resp["results"], case["want"], flags):
            assert not f & 1
            r = inp["resource"]
            assert have["resource"] == {"id": r.get("id", ""), "kind": r.get("kind", ""), "policyVersion": r.get("policyVersion", ""),
                                        "scope": r.get("scope", "")}
            assert have["actions"] == want["actions"], case["name"]
            if not include_meta:
                assert have["meta"] is None
                continue
            for a, m in want["meta"].items():
                assert have["meta"]["actions"][a] == m, (case["name"], a)
            if want["hasMeta"]:
                assert sorted(have["meta"]["effectiveDerivedRoles"]) == sorted(want["effectiveDerivedRoles"] or [])


@pytest.mark.parametrize("seed", range(4))
def test_response_assembly_matches_python_on_fuzzed_requests(seed):
    import copy

    import hostsim_api
    from cerbos_amd import capi
    rng = np.random.default_rng(20_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    inputs = _requests(rng, 300)
    inputs[5]["actions"] = ["view", "edit", "view", "edit", "view"]
    it, ins, batch = _same_as_check_inputs(lt, inputs)
    batch.actions_per_request = [list(i["actions"]) for i in ins]
    req = wire.encode_check_resources_request(_request_of(ins))
    res = hostsim_api.check(lt, batch, 1_700_000_000_000_000_000, capi.F_WANT_DERIVED_ROLES, device_order=True)
    raw, flags = it.assemble_response_pb(batch, res, req)
    resp = wire.decode_check_resources_response(raw)
    want, bad = HostSimEvaluator(lt, Conf()).assemble(ins, batch, copy.copy(res).to_input_order(batch), "default", allow_unsupported=True)
    assert sorted(bad) == [i for i, f in enumerate(flags) if f & 1]
    for have, w in zip(resp["results"], want):
        assert have["actions"] == {a: e["effect"] for a, e in w["actions"].items()}
        assert have["meta"]["actions"] == {a: {"matchedPolicy": e["policy"], "matchedScope": e["scope"]} for a, e in w["actions"].items()}
        assert have["meta"]["effectiveDerivedRoles"] == w["effectiveDerivedRoles"]


def test_mutated_requests_never_crash():
    """The same for corrupted CheckResourcesRequest bytes (and corrupted AuxData)."""
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    cases = load_json("server_check_cases.json")
    reqs = [wire.encode_check_resources_request(_request_of(c["inputs"])) for c in cases]
    aux = wire.encode_aux_data({"jwt": {"iss": "x", "aud": ["a", "b"], "n": {"k": [1, 2]}}, "jwts": {"t": {"claims": {"aud": ["a"]}}}})
    rng = np.random.default_rng(11)
    ok = bad = 0
    for trial in range(1200):
        m = bytearray(reqs[int(rng.integers(0, len(reqs)))])
        a = bytearray(aux)
        for buf in (m, a) if trial % 3 == 0 else (m,):
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(0, len(buf)))
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    buf[pos] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1:
                    del buf[pos:pos + int(rng.integers(1, 9))]
                else:
                    buf[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 6)), dtype=np.uint8))
        try:
            b = it.flatten_request_pb(bytes(m), bytes(a))
            assert b.tuple_action.size == b.n_tuples
            ok += 1
        except IngestError:
            bad += 1
    assert ok > 100 and bad > 100, (ok, bad)


def test_corrupted_table_images_are_refused_or_harmless():
    """Bit flips in the image's header / section table / anywhere: cbi_table_open refuses it or opens something
    that still flattens without touching memory outside the image (run under ASan/UBSan when changing the parser)."""
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    inputs = [i for c in load_json("engine_cases.json")[:5] for i in c["inputs"]]
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    rng = np.random.default_rng(1)
    opened = refused = 0
    for trial in range(800):
        b = bytearray(lt.blob)
        region = 32 + 32 * 27 if trial % 2 else len(b)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, region))] ^= 1 << int(rng.integers(0, 8))
        try:
            it = IngestTable(bytes(b))
        except IngestError:
            refused += 1
            continue
        opened += 1
        try:
            it.flatten_pb(data, off)
        except IngestError:
            pass
        it.close()
    assert opened > 100 and refused > 50, (opened, refused)


def test_check_request_pb_with_jwt_claims():
    """HipEvaluator.check_request_pb incl. the AuxData side channel: the verify vectors that carry JWT claims."""
    lt = lower_rule_table(store_rule_table(), {})
    ev = HostSimEvaluator(lt, Conf())
    n = 0
    for v in load_json("verify_vectors.json"):
        if "auxData" not in v["input"] or v["globals"]:
            continue
        from helpers import rfc3339_ns
        req = wire.encode_check_resources_request(_request_of([v["input"]], include_meta=False))
        raw, flags = ev.check_request_pb(req, wire.encode_aux_data(v["input"]["auxData"]),
                                         now_ns=rfc3339_ns(v["now"]) if v["now"] else 1_700_000_000_000_000_000,
                                         lenient_scope_search=v["lenient"], strict_evaluation=v["strict"],
                                         default_policy_version=v["defaultPolicyVersion"], default_scope=v["defaultScope"])
        assert not flags[0] & 1
        assert wire.decode_check_resources_response(raw)["results"][0]["actions"] == v["want"], (v["suite"], v["test"])
        n += 1
    assert n >= 20
