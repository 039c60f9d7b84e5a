"""Protobuf wire encoding of ``cerbos.engine.v1.CheckInput`` from its JSON-shaped dict.

What a Go caller already holds (``proto.Marshal(checkInput)``) and what ``libcerbos_ingest.so`` consumes;
here so that tests, the bench and Python callers can produce the same bytes without a protobuf runtime.
Field numbers: api/public/cerbos/engine/v1/engine.proto:130-200 (CheckInput 1 request_id, 2 resource,
3 principal, 4 actions, 5 aux_data; Resource 1 kind, 2 policy_version, 3 id, 4 attr, 5 scope; Principal 1 id,
2 policy_version, 3 roles, 4 attr, 5 scope; AuxData 1 jwt) and google/protobuf/struct.proto (Value 1 null,
2 number, 3 string, 4 bool, 5 struct, 6 list).
"""
from __future__ import annotations

import struct

import numpy as np


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:
    return _varint(field << 3 | 2) + _varint(len(payload)) + payload


def _string(field: int, s: str) -> bytes:
    return _ld(field, s.encode("utf-8")) if s else b""      # proto3: default values are not emitted


def encode_value(v) -> bytes:
    if v is None:
        return _varint(1 << 3 | 0) + b"\0"
    if isinstance(v, bool):
        return _varint(4 << 3 | 0) + (b"\1" if v else b"\0")
    if isinstance(v, (int, float)):
        return _varint(2 << 3 | 1) + struct.pack("<d", float(v))
    if isinstance(v, str):
        return _ld(3, v.encode("utf-8"))                    # a oneof member is emitted even when empty
    if isinstance(v, (list, tuple)):
        return _ld(6, b"".join(_ld(1, encode_value(x)) for x in v))
    if isinstance(v, dict):
        return _ld(5, encode_map(1, v))
    raise TypeError("unsupported attribute value %r" % (v,))


def encode_map(field: int, m: dict) -> bytes:
    """map<string, google.protobuf.Value> as repeated entries {1: key, 2: value}, in dict order."""
    return b"".join(_ld(field, _ld(1, str(k).encode("utf-8")) + _ld(2, encode_value(x))) for k, x in m.items())


def encode_resource(r: dict) -> bytes:
    return (_string(1, r.get("kind", "") or "") + _string(2, r.get("policyVersion", "") or "")
            + _string(3, r.get("id", "") or "") + encode_map(4, r.get("attr") or {}) + _string(5, r.get("scope", "") or ""))


def encode_principal(p: dict) -> bytes:
    return (_string(1, p.get("id", "") or "") + _string(2, p.get("policyVersion", "") or "")
            + b"".join(_ld(3, str(x).encode("utf-8")) for x in (p.get("roles") or []))
            + encode_map(4, p.get("attr") or {}) + _string(5, p.get("scope", "") or ""))


def encode_aux_data(aux: dict) -> bytes:
    """cerbos.engine.v1.AuxData: 1 jwt map<string, Value>, 2 jwts map<string, JWT{1 claims}>."""
    body = encode_map(1, aux.get("jwt") or {})
    for name, jwt in (aux.get("jwts") or {}).items():
        body += _ld(2, _ld(1, str(name).encode("utf-8")) + _ld(2, encode_map(1, (jwt or {}).get("claims") or {})))
    return body


def encode_check_input(inp: dict) -> bytes:
    aux = inp.get("auxData") or {}
    out = _string(1, inp.get("requestId", "") or "")
    out += _ld(2, encode_resource(inp.get("resource") or {}))
    out += _ld(3, encode_principal(inp.get("principal") or {}))
    out += b"".join(_ld(4, str(a).encode("utf-8")) for a in (inp.get("actions") or []))
    if aux.get("jwt") or aux.get("jwts"):
        out += _ld(5, encode_aux_data(aux))
    return out


def encode_check_resources_request(req: dict) -> bytes:
    """cerbos.request.v1.CheckResourcesRequest (request.proto:222-273): 1 request_id, 2 include_meta, 3 principal,
    4 resources {1 actions, 2 resource}.  (5 aux_data carries a JWT *token*; the claims travel as engine AuxData.)"""
    out = _string(1, req.get("requestId", "") or "")
    if req.get("includeMeta"):
        out += _varint(2 << 3 | 0) + b"\1"
    out += _ld(3, encode_principal(req.get("principal") or {}))
    for e in req.get("resources") or []:
        out += _ld(4, b"".join(_ld(1, str(a).encode("utf-8")) for a in (e.get("actions") or [])) + _ld(2, encode_resource(e.get("resource") or {})))
    return out


def pack_messages(msgs):
    """[bytes] -> (one contiguous uint8 array, uint64 offsets[n + 1])."""
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    if msgs:
        off[1:] = np.cumsum([len(m) for m in msgs])
    return np.frombuffer(b"".join(msgs), dtype=np.uint8), off


# ---- decoding of enginev1.CheckOutput (engine.proto:152-166), for tests and Python callers -----------------
def _fields(buf: bytes):
    i, n = 0, len(buf)
    while i < n:
        key = shift = 0
        while True:
            b = buf[i]; i += 1
            key |= (b & 0x7F) << shift; shift += 7
            if not b & 0x80:
                break
        num, wt = key >> 3, key & 7
        if wt == 0:
            v = shift = 0
            while True:
                b = buf[i]; i += 1
                v |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            yield num, v
        elif wt == 2:
            ln = shift = 0
            while True:
                b = buf[i]; i += 1
                ln |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            yield num, buf[i:i + ln]
            i += ln
        else:
            raise ValueError("unexpected wire type %d" % wt)


_EFFECTS = {0: "EFFECT_UNSPECIFIED", 1: "EFFECT_ALLOW", 2: "EFFECT_DENY", 3: "EFFECT_NO_MATCH"}


def decode_value(buf: bytes):
    """google.protobuf.Value -> JSON (the last field wins; an empty message is null)."""
    import struct
    out = None
    i, n = 0, len(buf)
    while i < n:
        key = buf[i]; i += 1
        num, wt = key >> 3, key & 7
        if wt == 1:
            out = struct.unpack_from("<d", buf, i)[0]; i += 8
        elif wt == 0:
            v = shift = 0
            while True:
                b = buf[i]; i += 1
                v |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            out = None if num == 1 else bool(v)
        else:
            ln = shift = 0
            while True:
                b = buf[i]; i += 1
                ln |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            body = buf[i:i + ln]; i += ln
            if num == 3:
                out = body.decode("utf-8")
            elif num == 6:
                out = [decode_value(v2) for n2, v2 in _fields(body) if n2 == 1]
            elif num == 5:
                out = {}
                for n2, ent in _fields(body):
                    if n2 == 1:
                        k, val = "", None
                        for n3, v3 in _fields(ent):
                            if n3 == 1:
                                k = v3.decode("utf-8")
                            elif n3 == 2:
                                val = decode_value(v3)
                        out[k] = val
    return out


def decode_check_output(buf: bytes) -> dict:
    out = {"requestId": "", "resourceId": "", "actions": {}, "effectiveDerivedRoles": []}
    for num, v in _fields(buf):
        if num == 6:     # OutputEntry {src 1, val 2, action 3, error 4}
            e = {}
            has_val = False
            for n2, v2 in _fields(v):
                if n2 == 1:
                    e["src"] = v2.decode("utf-8")
                elif n2 == 2:
                    e["val"] = decode_value(v2); has_val = True
                elif n2 == 3:
                    e["action"] = v2.decode("utf-8")
                elif n2 == 4:
                    e["error"] = v2.decode("utf-8")
            if not has_val and "error" not in e:
                e["val"] = None
            out.setdefault("outputs", []).append(e)
            continue
        if num == 7:     # EvaluationError {cel_error 1 {expression 1, message 2}}
            ce = {"expression": "", "message": ""}
            for n2, v2 in _fields(v):
                if n2 == 1:
                    for n3, v3 in _fields(v2):
                        ce["expression" if n3 == 1 else "message"] = v3.decode("utf-8")
            out.setdefault("evaluationErrors", []).append({"celError": ce})
            continue
        if num == 1:
            out["requestId"] = v.decode("utf-8")
        elif num == 2:
            out["resourceId"] = v.decode("utf-8")
        elif num == 3:
            key, eff = "", {"effect": _EFFECTS[0], "policy": "", "scope": ""}
            for n2, v2 in _fields(v):
                if n2 == 1:
                    key = v2.decode("utf-8")
                elif n2 == 2:
                    for n3, v3 in _fields(v2):
                        if n3 == 1:
                            eff["effect"] = _EFFECTS[v3]
                        elif n3 == 2:
                            eff["policy"] = v3.decode("utf-8")
                        elif n3 == 3:
                            eff["scope"] = v3.decode("utf-8")
            out["actions"][key] = eff
        elif num == 4:
            out["effectiveDerivedRoles"].append(v.decode("utf-8"))
    return out


def decode_check_resources_response(buf: bytes) -> dict:
    """cerbos.response.v1.CheckResourcesResponse (response.proto:187-300)."""
    out = {"requestId": "", "results": []}
    for num, v in _fields(buf):
        if num == 1:
            out["requestId"] = v.decode("utf-8")
        elif num == 2:
            entry = {"resource": {"id": "", "kind": "", "policyVersion": "", "scope": ""}, "actions": {}, "meta": None}
            for n2, v2 in _fields(v):
                if n2 == 1:
                    for n3, v3 in _fields(v2):
                        entry["resource"][{1: "id", 2: "kind", 3: "policyVersion", 4: "scope"}[n3]] = v3.decode("utf-8")
                elif n2 == 2:
                    key, eff = "", _EFFECTS[0]
                    for n3, v3 in _fields(v2):
                        if n3 == 1:
                            key = v3.decode("utf-8")
                        elif n3 == 2:
                            eff = _EFFECTS[v3]
                    entry["actions"][key] = eff
                elif n2 == 5:    # OutputEntry {src 1, val 2, action 3, error 4}
                    e2 = {}
                    for n3, v3 in _fields(v2):
                        if n3 == 1:
                            e2["src"] = v3.decode("utf-8")
                        elif n3 == 2:
                            e2["val"] = decode_value(v3)
                        elif n3 == 3:
                            e2["action"] = v3.decode("utf-8")
                        elif n3 == 4:
                            e2["error"] = v3.decode("utf-8")
                    if "val" not in e2 and "error" not in e2:
                        e2["val"] = None
                    entry.setdefault("outputs", []).append(e2)
                elif n2 == 4:
                    meta = {"actions": {}, "effectiveDerivedRoles": []}
                    for n3, v3 in _fields(v2):
                        if n3 == 1:
                            key, em = "", {"matchedPolicy": "", "matchedScope": ""}
                            for n4, v4 in _fields(v3):
                                if n4 == 1:
                                    key = v4.decode("utf-8")
                                elif n4 == 2:
                                    for n5, v5 in _fields(v4):
                                        em[{1: "matchedPolicy", 2: "matchedScope"}[n5]] = v5.decode("utf-8")
                            meta["actions"][key] = em
                        elif n3 == 2:
                            meta["effectiveDerivedRoles"].append(v3.decode("utf-8"))
                    entry["meta"] = meta
            out["results"].append(entry)
    return out


# ---- PlanResources (engine.proto:20-128): input 1 request_id, 2 action (deprecated), 3 principal, 4 resource {1 kind, 2 attr,
# 3 policy_version, 4 scope}, 5 aux_data, 6 include_meta, 7 actions; output 1 request_id, 2 action, 3 kind, 4 policy_version, 5 scope,
# 6 filter {1 kind, 2 condition}, 7 filter_debug, 9 actions, 10 matched_scopes, 11 evaluation_errors: EvaluationError {1 cel_error
# {1 expression, 2 message}} (engine.proto:127, 168-176 - the message a CheckOutput's field 7 repeats);
# Operand: 1 value, 2 expression {1 operator, 2 operands}, 3 variable
_FILTER_KINDS = {"KIND_UNSPECIFIED": 0, "KIND_ALWAYS_ALLOWED": 1, "KIND_ALWAYS_DENIED": 2, "KIND_CONDITIONAL": 3}


def _decode_map(entries):
    out = {}
    for ent in entries:
        k, val = "", None
        for n3, v3 in _fields(ent):
            if n3 == 1:
                k = v3.decode("utf-8")
            elif n3 == 2:
                val = decode_value(v3)
        out[k] = val
    return out


def _decode_principal(buf):
    p = {"id": "", "policyVersion": "", "roles": [], "attr": {}, "scope": ""}
    attrs = []
    for n, v in _fields(buf):
        if n == 1:
            p["id"] = v.decode("utf-8")
        elif n == 2:
            p["policyVersion"] = v.decode("utf-8")
        elif n == 3:
            p["roles"].append(v.decode("utf-8"))
        elif n == 4:
            attrs.append(v)
        elif n == 5:
            p["scope"] = v.decode("utf-8")
    p["attr"] = _decode_map(attrs)
    return p


def encode_plan_resources_input(inp: dict) -> bytes:
    r = inp.get("resource") or {}
    res = (_string(1, r.get("kind", "") or "") + encode_map(2, r.get("attr") or {}) + _string(3, r.get("policyVersion", "") or "")
           + _string(4, r.get("scope", "") or ""))
    out = _string(1, inp.get("requestId", "") or "") + _string(2, inp.get("action", "") or "")
    out += _ld(3, encode_principal(inp.get("principal") or {})) + _ld(4, res)
    aux = inp.get("auxData") or {}
    if aux.get("jwt") or aux.get("jwts"):
        out += _ld(5, encode_aux_data(aux))
    if inp.get("includeMeta"):
        out += _varint(6 << 3 | 0) + b"\1"
    out += b"".join(_ld(7, str(a).encode("utf-8")) for a in (inp.get("actions") or []))
    return out


def decode_plan_resources_input(buf: bytes) -> dict:
    inp = {"requestId": "", "actions": [], "principal": {}, "resource": {}, "auxData": None, "includeMeta": False}
    action = ""
    for n, v in _fields(buf):
        if n == 1:
            inp["requestId"] = v.decode("utf-8")
        elif n == 2:
            action = v.decode("utf-8")
        elif n == 3:
            inp["principal"] = _decode_principal(v)
        elif n == 4:
            r, attrs = {"kind": "", "policyVersion": "", "scope": ""}, []
            for n2, v2 in _fields(v):
                if n2 == 1:
                    r["kind"] = v2.decode("utf-8")
                elif n2 == 2:
                    attrs.append(v2)
                elif n2 == 3:
                    r["policyVersion"] = v2.decode("utf-8")
                elif n2 == 4:
                    r["scope"] = v2.decode("utf-8")
            r["attr"] = _decode_map(attrs)
            inp["resource"] = r
        elif n == 5:
            aux = {"jwt": _decode_map([v2 for n2, v2 in _fields(v) if n2 == 1]), "jwts": {}}
            for n2, v2 in _fields(v):
                if n2 == 2:          # map<string, JWT {1 claims}>
                    name, claims = "", {}
                    for n3, v3 in _fields(v2):
                        if n3 == 1:
                            name = v3.decode("utf-8")
                        elif n3 == 2:
                            claims = _decode_map([v4 for n4, v4 in _fields(v3) if n4 == 1])
                    aux["jwts"][name] = {"claims": claims}
            inp["auxData"] = aux
        elif n == 6:
            inp["includeMeta"] = bool(v)
        elif n == 7:
            inp["actions"].append(v.decode("utf-8"))
    inp["action"] = action      # (deprecated field 2: kept apart - MkPlanResourcesOutput copies both as they came, planner.go:113-125)
    return inp


def _encode_operand(op: dict) -> bytes:
    if "expression" in op:
        e = op["expression"]
        body = _string(1, e["operator"]) + b"".join(_ld(2, _encode_operand(o)) for o in e.get("operands") or [])
        return _ld(2, body)
    if "variable" in op:
        return _ld(3, op["variable"].encode("utf-8"))
    return _ld(1, encode_value(op["value"]))


def encode_plan_resources_output(out: dict) -> bytes:
    f = out["filter"]
    fb = _varint(1 << 3 | 0) + _varint(_FILTER_KINDS[f["kind"]])
    if f.get("condition") is not None:
        fb += _ld(2, _encode_operand(f["condition"]))
    acts = out.get("actions") or []
    b = _string(1, out.get("requestId", "")) + _string(2, out.get("action", "")) + _string(3, out.get("kind", "")) + _string(4, out.get("policyVersion", "")) + _string(5, out.get("scope", ""))
    b += _ld(6, fb) + _string(7, out.get("filterDebug", ""))
    b += b"".join(_ld(9, a.encode("utf-8")) for a in acts)
    for k, v in (out.get("matchedScopes") or {}).items():
        b += _ld(10, _ld(1, k.encode("utf-8")) + _string(2, v))
    for e in out.get("evaluationErrors") or []:
        ce = e.get("celError") or {}
        b += _ld(11, _ld(1, _string(1, ce.get("expression", "")) + _string(2, ce.get("message", ""))))
    return b


def _decode_operand(buf: bytes) -> dict:
    for n, v in _fields(buf):
        if n == 1:
            return {"value": decode_value(v)}
        if n == 3:
            return {"variable": v.decode("utf-8")}
        if n == 2:
            e = {"operator": "", "operands": []}
            for n2, v2 in _fields(v):
                if n2 == 1:
                    e["operator"] = v2.decode("utf-8")
                elif n2 == 2:
                    e["operands"].append(_decode_operand(v2))
            return {"expression": e}
    return {"value": None}


def decode_plan_resources_output(buf: bytes) -> dict:
    kinds = {v: k for k, v in _FILTER_KINDS.items()}
    out = {"requestId": "", "action": "", "kind": "", "policyVersion": "", "scope": "", "filter": {"kind": "KIND_UNSPECIFIED"}, "filterDebug": "",
           "actions": [], "matchedScopes": {}, "evaluationErrors": []}
    for n, v in _fields(buf):
        if n in (1, 2, 3, 4, 5, 7):
            out[{1: "requestId", 2: "action", 3: "kind", 4: "policyVersion", 5: "scope", 7: "filterDebug"}[n]] = v.decode("utf-8")
        elif n == 6:
            f = {"kind": "KIND_UNSPECIFIED"}
            for n2, v2 in _fields(v):
                if n2 == 1:
                    f["kind"] = kinds[v2]
                elif n2 == 2:
                    f["condition"] = _decode_operand(v2)
            out["filter"] = f
        elif n == 9:
            out["actions"].append(v.decode("utf-8"))
        elif n == 10:
            k = val = ""
            for n2, v2 in _fields(v):
                if n2 == 1:
                    k = v2.decode("utf-8")
                elif n2 == 2:
                    val = v2.decode("utf-8")
            out["matchedScopes"][k] = val
        elif n == 11:
            ce = {"expression": "", "message": ""}
            for n2, v2 in _fields(v):
                if n2 == 1:
                    for n3, v3 in _fields(v2):
                        ce["expression" if n3 == 1 else "message"] = v3.decode("utf-8")
            out["evaluationErrors"].append({"celError": ce})
    return out
