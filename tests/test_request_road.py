"""The device road for serialized CheckResourcesRequests (cerbos_amd/csrc/cbh_wire_req.h: count + split kernels in front of the
device flattener) on the simulator: the CheckInputs the split makes must be the ones svc.CheckResources would build
(cerbos_svc.go:274-288), the batch flattened from them the one the host road (cbi_flatten_request_pb) makes, and the answers the
oracle's."""
import ctypes as C

import numpy as np
import pytest

import hostsim_api
import wire_device_util as wdu
from cerbos_amd import capi, wire
from cerbos_amd.ingest import IngestError, IngestTable
from cerbos_amd.lower.blob import lower_rule_table
from helpers import load_json, store_rule_table
from test_hostsim_golden import GLOBALS

NOW = 1_700_000_000_000_000_000


def sim_split(requests, aux=None):
    """[serialized CheckResourcesRequest] (+ per request serialized engine AuxData or None) -> (messages, first_input, flags) or the
    index of the first malformed request."""
    lib = hostsim_api.lib()
    lib.hostsim_wire_split_requests.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p] + [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(C.c_uint32)]
    lib.hostsim_wire_split_requests.restype = C.c_longlong
    data, off = wire.pack_messages(list(requests))
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    a_data = a_off = None
    if aux is not None:
        a_data, a_off = wire.pack_messages([x or b"" for x in aux])
        a_data = np.concatenate([a_data, np.zeros(8, np.uint8)])
    msg, moff, first, flags = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    bad = C.c_uint32()
    n = lib.hostsim_wire_split_requests(data.ctypes.data, off.ctypes.data, len(requests), a_data.ctypes.data if a_data is not None else None,
                                        a_off.ctypes.data if a_off is not None else None, C.byref(msg), C.byref(moff), C.byref(first), C.byref(flags), C.byref(bad))
    if n < 0:
        return int(bad.value)
    nr = len(requests)
    moff = np.ctypeslib.as_array(C.cast(moff, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    first = np.ctypeslib.as_array(C.cast(first, C.POINTER(C.c_uint32)), shape=(nr + 1,)).copy()
    flags = np.ctypeslib.as_array(C.cast(flags, C.POINTER(C.c_uint8)), shape=(max(nr, 1),)).copy()[:nr]
    total = int(moff[n])
    raw = np.ctypeslib.as_array(C.cast(msg, C.POINTER(C.c_uint8)), shape=(total + 1,)).copy()
    assert raw[total] == 0xEE, "the split wrote past the bytes the count announced"
    raw = raw[:total].tobytes()
    return [raw[int(moff[i]):int(moff[i + 1])] for i in range(n)], first, flags


def _request_of(inputs, include_meta=False):
    return {"requestId": inputs[0].get("requestId", ""), "includeMeta": include_meta, "principal": inputs[0]["principal"],
            "resources": [{"actions": i["actions"], "resource": i["resource"]} for i in inputs]}


def _as_built_by_the_service(inputs, aux=None):
    """cerbos_svc.go:274-288: the request's id and principal on every entry"""
    out = []
    for i in inputs:
        d = dict(i, principal=inputs[0]["principal"], requestId=inputs[0].get("requestId", ""))
        d.pop("auxData", None)
        if aux:
            d["auxData"] = aux
        out.append(d)
    return out


def _cases():
    return [c["inputs"] for c in load_json("server_check_cases.json")] + [c["inputs"] for c in load_json("engine_cases.json") if c["inputs"]]


def test_split_builds_the_check_inputs_of_the_service():
    groups = _cases()
    reqs = [wire.encode_check_resources_request(_request_of(g, include_meta=(k % 2 == 0))) for k, g in enumerate(groups)]
    msgs, first, flags = sim_split(reqs)
    assert list(np.diff(first)) == [len(g) for g in groups]
    assert [bool(f & 1) for f in flags] == [k % 2 == 0 for k in range(len(groups))]
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    want_inputs = [i for g in groups for i in _as_built_by_the_service(g)]
    want = it.flatten_pb(*wire.pack_messages([wire.encode_check_input(i) for i in want_inputs]), sort=0)
    have = it.flatten_pb(*wire.pack_messages(msgs), sort=0)
    for name in ("req_u32", "roles", "tuple_action", "col_tag", "col_val", "heap_tag", "heap_val", "str_off", "str_bytes"):
        assert np.array_equal(getattr(have, name), getattr(want, name)), name
    # ... and message by message the service's fields in the engine's numbering: 1 request id, 2 resource, 3 principal, 4 actions
    for m, i in zip(msgs, want_inputs):
        fields = list(wire._fields(m))
        assert [v for n, v in fields if n == 4] == [a.encode() for a in i["actions"]]
        assert [v for n, v in fields if n == 3] == [wire.encode_principal(i["principal"])]
        assert [v for n, v in fields if n == 2] == [wire.encode_resource(i["resource"])]
        assert [v for n, v in fields if n == 1] == ([i["requestId"].encode()] if i.get("requestId") else [])


def test_engine_aux_data_rides_on_every_entry():
    jwt_inputs = [v["input"] for v in load_json("verify_vectors.json") if "auxData" in v["input"]]
    plain = _cases()[0]
    aux = jwt_inputs[0]["auxData"]
    reqs = [wire.encode_check_resources_request(_request_of(plain)), wire.encode_check_resources_request(_request_of(jwt_inputs[:1])),
            wire.encode_check_resources_request(_request_of(plain))]
    msgs, first, _ = sim_split(reqs, aux=[None, wire.encode_aux_data(aux), None])
    assert list(first) == [0, len(plain), len(plain) + 1, 2 * len(plain) + 1]
    for k, m in enumerate(msgs):
        got = [v for n, v in wire._fields(m) if n == 5]
        assert got == ([wire.encode_aux_data(aux)] if k == len(plain) else [])
    assert msgs[len(plain)] == wire.encode_check_input(_as_built_by_the_service(jwt_inputs[:1], aux)[0]) or \
        sorted(wire._fields(msgs[len(plain)])) == sorted(wire._fields(wire.encode_check_input(_as_built_by_the_service(jwt_inputs[:1], aux)[0])))


def _decode_check_input(buf):
    """engine.proto CheckInput as a dict, read the protobuf way: a message field named more than once is the merge of its occurrences
    (= its occurrences' bytes back to back), a scalar named more than once its last occurrence"""
    parts = {2: b"", 3: b""}
    out = {"requestId": "", "actions": []}
    for n, v in wire._fields(buf):
        if n in parts:
            parts[n] += v
        elif n == 1:
            out["requestId"] = v.decode()
        elif n == 4:
            out["actions"].append(v.decode())
    out["principal"] = wire._decode_principal(parts[3])
    r = {"kind": "", "policyVersion": "", "id": "", "scope": ""}
    attrs = []
    for n, v in wire._fields(parts[2]):
        if n == 4:
            attrs.append(v)
        elif n in (1, 2, 3, 5):
            r[{1: "kind", 2: "policyVersion", 3: "id", 5: "scope"}[n]] = v.decode()
    r["attr"] = wire._decode_map(attrs)
    out["resource"] = r
    return out


def test_unusual_but_valid_encodings():
    """Fields in any order, repeated, unknown fields, empty requests, entries without a resource.  A scalar named twice: the last
    wins.  A MESSAGE field named twice (the principal, an entry's resource): its occurrences merge, as proto.Unmarshal merged them for
    the server that validated and logged the request - the CheckInput carries them back to back, which parses as the merge."""
    p1, p2 = wire.encode_principal({"id": "a", "roles": ["x"]}), wire.encode_principal({"id": "b", "roles": ["y"]})
    r1 = wire.encode_resource({"kind": "k", "id": "1"})
    r2 = wire.encode_resource({"kind": "k", "id": "2"})
    ld = wire._ld
    entry = ld(1, b"view") + ld(2, r1) + ld(1, b"edit") + ld(2, r2)          # the resources merge (id: the later one), the actions accumulate
    unknown = wire._varint(9 << 3 | 0) + b"\x05" + ld(12, b"zzz") + wire._varint(10 << 3 | 1) + bytes(8) + wire._varint(11 << 3 | 5) + bytes(4)
    req = ld(4, entry) + unknown + ld(3, p1) + ld(1, b"first") + ld(4, ld(1, b"only-actions")) + ld(3, p2) + ld(1, b"second") + ld(4, b"") + ld(6, b"ctx")
    msgs, first, flags = sim_split([b"", req, ld(3, p1)])
    assert list(first) == [0, 0, 3, 3] and not flags.any()
    f0 = list(wire._fields(msgs[0]))
    assert f0 == [(1, b"second"), (2, r1 + r2), (3, p1 + p2), (4, b"view"), (4, b"edit")]
    assert list(wire._fields(msgs[1])) == [(1, b"second"), (3, p1 + p2), (4, b"only-actions")]
    assert list(wire._fields(msgs[2])) == [(1, b"second"), (3, p1 + p2)]
    merged = _decode_check_input(msgs[0])
    assert merged["principal"]["id"] == "b" and merged["principal"]["roles"] == ["x", "y"] and merged["resource"]["id"] == "2"
    # the host road reads the same request the same way
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    host = it.flatten_request_pb(req, sort=0)
    dev = it.flatten_pb(*wire.pack_messages(msgs), sort=0)
    for name in ("req_u32", "roles", "tuple_action", "col_tag", "col_val"):
        assert np.array_equal(getattr(host, name), getattr(dev, name)), name


def test_a_message_field_named_twice_is_merged_not_replaced():
    """Requests of the service cases re-encoded so that the principal and every entry's resource arrive in TWO pieces (identity and roles
    first, attributes - one of them overridden - later): both roads must read them as the canonical request they merge to."""
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    ld = wire._ld
    n = 0
    for group in _cases()[:10]:
        p = group[0]["principal"]
        attr = dict(p.get("attr") or {})
        first_half = wire.encode_principal({"id": p["id"], "roles": p["roles"][:1], "policyVersion": p.get("policyVersion", ""),
                                            "attr": dict({k: "overridden" for k in list(attr)[:1]})})
        second_half = wire.encode_principal({"roles": p["roles"][1:], "attr": attr, "scope": p.get("scope", "")})
        entries_split, entries_canon = b"", b""
        for i in group:
            r = i["resource"]
            ra = dict(r.get("attr") or {})
            ka, kb = dict(list(ra.items())[: len(ra) // 2]), dict(list(ra.items())[len(ra) // 2:])
            a = wire.encode_resource({"kind": r["kind"], "id": "to-be-replaced", "attr": ka})
            b = wire.encode_resource({"id": r["id"], "policyVersion": r.get("policyVersion", ""), "scope": r.get("scope", ""), "attr": kb})
            acts = b"".join(ld(1, x.encode()) for x in i["actions"])
            entries_split += ld(4, ld(2, a) + acts + ld(2, b))
            entries_canon += ld(4, acts + ld(2, wire.encode_resource(r)))
        rid = ld(1, group[0].get("requestId", "").encode()) if group[0].get("requestId") else b""
        split = rid + ld(3, first_half) + entries_split + ld(3, second_half)
        canon = rid + ld(3, wire.encode_principal(p)) + entries_canon
        (m_split, _, _), (m_canon, _, _) = sim_split([split]), sim_split([canon])
        assert len(m_split) == len(m_canon) == len(group)
        for x, y in zip(m_split, m_canon):
            dx, dy = _decode_check_input(x), _decode_check_input(y)
            assert dx == dy, (dx, dy)
        h_split, h_canon = it.flatten_request_pb(split, sort=0), it.flatten_request_pb(canon, sort=0)
        d_split = it.flatten_pb(*wire.pack_messages(m_split), sort=0)
        for name in ("req_u32", "roles", "tuple_action", "col_tag"):
            assert np.array_equal(getattr(h_split, name), getattr(h_canon, name)), name
            assert np.array_equal(getattr(h_split, name), getattr(d_split, name)), name
        n += len(group)
    assert n > 10


def test_corrupted_requests_are_refused_like_on_the_host_road():
    """Truncations and byte flips: the split refuses exactly the requests whose own framing (top level, entries) is broken - what
    split_request (cbh_ingest.cpp) refuses or what its entry walk would; it never writes outside what it announced (sim_split checks)."""
    groups = _cases()[:12]
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    it = IngestTable(lt.blob)
    rng = np.random.default_rng(11)
    refused = accepted = 0
    for trial in range(600):
        g = groups[trial % len(groups)]
        m = bytearray(wire.encode_check_resources_request(_request_of(g, include_meta=bool(trial & 1))))
        kind = trial % 3
        pos = int(rng.integers(0, len(m)))
        if kind == 0:
            del m[pos:]
        elif kind == 1:
            m[pos] ^= 1 << int(rng.integers(0, 8))
        else:
            m[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 5)), dtype=np.uint8))
        good = wire.encode_check_resources_request(_request_of(groups[(trial + 1) % len(groups)]))
        got = sim_split([good, bytes(m), good])
        try:
            host = it.flatten_request_pb(bytes(m), sort=0)
            host_ok = True
        except IngestError:
            host_ok = False
        if isinstance(got, int):
            assert got == 1
            assert not host_ok, "the device refuses a request the host road reads"
            refused += 1
            continue
        accepted += 1
        msgs, first, _ = got
        mine = msgs[int(first[1]):int(first[2])]
        # framing fine; what is inside an entry's resource / the principal is the flattener's to judge - the same way on both roads
        try:
            dev = it.flatten_pb(*wire.pack_messages(mine), sort=0)
            dev_ok = True
        except IngestError:
            dev_ok = False
        assert dev_ok == host_ok, trial
        if host_ok:
            for name in ("req_u32", "roles", "tuple_action", "col_tag", "col_val"):
                assert np.array_equal(getattr(host, name), getattr(dev, name)), (trial, name)
    assert refused > 50 and accepted > 50, (refused, accepted)


@pytest.mark.parametrize("case", load_json("server_check_cases.json"), ids=lambda c: c["name"])
def test_requests_down_the_device_road_on_the_simulator(case):
    """split -> device flattener -> decision kernels -> device assembler, all on the simulator: the reference's service-level cases."""
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    req = wire.encode_check_resources_request(_request_of(case["inputs"], True))
    msgs, first, flags = sim_split([req, req])
    assert flags.all() and list(first) == [0, len(case["inputs"]), 2 * len(case["inputs"])]
    data, off = wire.pack_messages(msgs)
    rc, wb = wdu.sim_flatten(lt, data, off)
    assert rc == 0
    res = hostsim_api.check(lt, wdu.to_batch(lt, wb), NOW, capi.F_WANT_DERIVED_ROLES, device_order=True)
    outs, oflags = wdu.sim_outputs(lt, res, len(msgs))
    for k, (inp, want) in enumerate(list(zip(case["inputs"], case["want"])) * 2):
        assert not oflags[k] & 1
        have = wire.decode_check_output(outs[k])
        assert have["resourceId"] == inp["resource"].get("id", "") and have["requestId"] == case["inputs"][0].get("requestId", "")
        assert {a: e["effect"] for a, e in have["actions"].items()} == want["actions"], case["name"]
        for a, m in want["meta"].items():
            assert have["actions"][a]["policy"] == m["matchedPolicy"] and have["actions"][a]["scope"] == m["matchedScope"]


def test_auxiliary_offsets_that_point_outside_are_refused():
    """[0, far beyond the bytes, ..]: the request is named, nothing outside the auxiliary bytes is read."""
    import ctypes as C
    lib = hostsim_api.lib()
    lib.hostsim_wire_split_requests.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p] + [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(C.c_uint32)]
    lib.hostsim_wire_split_requests.restype = C.c_longlong
    req = wire.encode_check_resources_request(_request_of(_cases()[0]))
    data, off = wire.pack_messages([req, req, req])
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    aux = np.zeros(64, np.uint8)
    aoff = np.array([0, 10, 1_000_000, 20], dtype=np.uint64)          # request 1's auxiliary data "ends" a megabyte away
    outs = [C.c_void_p() for _ in range(4)]
    bad = C.c_uint32()
    n = lib.hostsim_wire_split_requests(data.ctypes.data, off.ctypes.data, 3, aux.ctypes.data, aoff.ctypes.data, *[C.byref(o) for o in outs], C.byref(bad))
    assert n < 0 and bad.value == 1


def _trail_want(oracle, inputs, params):
    keys = set()
    for i in inputs:
        keys.update(oracle.check(i, params)["effectivePolicies"])
    return sorted(keys)


def test_every_request_s_audit_trail_beside_its_outputs(monkeypatch):
    """cbh_wire_check_requests_trail_pb's grouping on the simulator: one trail group per request (the one decision-log entry the server
    writes for a call), also when the flattener reorders the batch by route - against the oracle's union over the request's entries."""
    from cerbos_amd.engine import Conf
    from oracle.check import EvalParams, RuleTableOracle
    from test_hostsim_golden import HostSimEvaluator
    rt = store_rule_table()
    ev, oracle = HostSimEvaluator(lower_rule_table(rt, GLOBALS), Conf(globals_=GLOBALS)), RuleTableOracle(rt)
    params = EvalParams(globals_=GLOBALS, now_ns=NOW)
    cases = load_json("server_check_cases.json") + [c for c in load_json("engine_cases.json") if not c["wantError"]]
    groups = [c["inputs"] for c in cases if not any("auxData" in i for i in c["inputs"])] + [[]]
    reqs = [wire.encode_check_resources_request(_request_of(g)) if g else b"" for g in groups]
    for group_by_route in ("0", "1"):
        monkeypatch.setenv("CBH_WIRE_GROUP", group_by_route)
        outs, oflags, _, trails = ev.check_requests_pb(reqs, now_ns=NOW, audit_trail=True)
        plain, plain_flags, _ = ev.check_requests_pb(reqs, now_ns=NOW)
        assert outs == plain and np.array_equal(oflags, plain_flags)     # the trail's kernels decide - and flag - like the ordinary ones
        k = compared = 0
        for g, trail in zip(groups, trails):
            flagged = any(oflags[k + j] & 1 for j in range(len(g)))
            k += len(g)
            if flagged:
                continue
            assert trail == _trail_want(oracle, _as_built_by_the_service(g), params), g
            compared += 1
        assert compared > 40 and trails[-1] == []
