"""``runtimev1.RuleTable`` wire bytes <-> the rule-table dict the lowering consumes.

A deployed Cerbos does not hand policies to the engine, it hands over its compiled artefact: the
``runtimev1.RuleTable`` message (``api/private/cerbos/runtime/v1/runtime.proto:41-105``; produced by
``ruletable.NewProtoRuletable*`` / loaded by ``private/ruletable/ruletable.go:27-44`` and
``internal/ruletable/index/marshal.go``).  ``decode_rule_table`` turns those bytes into exactly the dict
``cerbos_amd.ruletable.build.build_rule_table`` produces from policy YAML - same rows in the same order, same side
tables - so ``cerbos_amd.lower.blob.lower_rule_table`` gives a byte-identical device image either way
(``tests/test_ruletable_proto.py``).  The index side tables the message does not carry (scope maps, scope
permissions, parent-role closure) are rebuilt by ``finish_rule_table`` as ``indexRules`` does at load
(``ruletable.go:798-834``).

Expressions: a ``runtimev1.Expr`` carries the source text (``original``) and cel-go's type-checked AST
(``checked``, ``google.api.expr.v1alpha1.CheckedExpr``).  The CEL front end here works from the source text -
the same text the policy author wrote and ``Expr.Original`` preserves - and ignores ``checked``.

``encode_rule_table`` is the inverse (what the Go side's ``proto.Marshal`` of its table would emit, field for
field); it exists for fixtures and round-trip tests - there is no Go toolchain here to produce the bytes - and for
hosts that want to ship a table between processes.  Module ids (map keys of ``meta`` / ``policy_derived_roles``)
are ``namer.GenModuleIDFromFQN`` = xxhash64 of the FQN (``internal/namer/namer.go:54-56``, ``util.HashStr``).

Protobuf wire format is hand-rolled (no protoc in the image): field numbers below are those of runtime.proto.
"""
from __future__ import annotations

import struct

from .build import KIND_PRINCIPAL, KIND_RESOURCE, finish_rule_table

_KIND_NUM = {KIND_PRINCIPAL: 3, KIND_RESOURCE: 4}          # policy.v1.Kind (policy.proto:19-27)
_KIND_NAME = {v: k for k, v in _KIND_NUM.items()}
_EFFECT_NUM = {"ALLOW": 1, "DENY": 2}                      # effect.v1.Effect
_EFFECT_NAME = {v: k for k, v in _EFFECT_NUM.items()}
_OPS = {"all": 1, "any": 2, "none": 3}                     # runtimev1.Condition oneof (runtime.proto:291-302)
_OP_NAMES = {v: k for k, v in _OPS.items()}


# ---- xxhash64 (util.HashStr) ---------------------------------------------------------------------------------
_P1, _P2, _P3, _P4, _P5 = 0x9E3779B185EBCA87, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x85EBCA77C2B2AE63, 0x27D4EB2F165667C5
_M = (1 << 64) - 1


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & _M


def _round(acc, lane):
    return (_rotl((acc + lane * _P2) & _M, 31) * _P1) & _M


def _merge(h, v):
    return ((h ^ _round(0, v)) * _P1 + _P4) & _M


def xxhash64(data: bytes, seed: int = 0) -> int:
    n, i = len(data), 0
    if n >= 32:
        v = [(seed + _P1 + _P2) & _M, (seed + _P2) & _M, seed & _M, (seed - _P1) & _M]
        while i + 32 <= n:
            for k in range(4):
                v[k] = _round(v[k], struct.unpack_from("<Q", data, i + 8 * k)[0])
            i += 32
        h = (_rotl(v[0], 1) + _rotl(v[1], 7) + _rotl(v[2], 12) + _rotl(v[3], 18)) & _M
        for k in range(4):
            h = _merge(h, v[k])
    else:
        h = (seed + _P5) & _M
    h = (h + n) & _M
    while i + 8 <= n:
        h = (_rotl(h ^ _round(0, struct.unpack_from("<Q", data, i)[0]), 27) * _P1 + _P4) & _M
        i += 8
    if i + 4 <= n:
        h = (_rotl(h ^ (struct.unpack_from("<I", data, i)[0] * _P1 & _M), 23) * _P2 + _P3) & _M
        i += 4
    while i < n:
        h = (_rotl(h ^ (data[i] * _P5 & _M), 11) * _P1) & _M
        i += 1
    h ^= h >> 33
    h = (h * _P2) & _M
    h ^= h >> 29
    h = (h * _P3) & _M
    h ^= h >> 32
    return h


def module_id(fqn: str) -> int:
    """namer.GenModuleIDFromFQN(fqn).RawValue()"""
    return xxhash64(fqn.encode("utf-8"))


# ---- wire helpers --------------------------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    out = bytearray()
    n &= _M
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _str(field: int, s: str) -> bytes:
    return _ld(field, s.encode("utf-8")) if s else b""


def _uint(field: int, v: int) -> bytes:
    return _varint(field << 3) + _varint(v) if v else b""


def _fields(buf):
    """Yields (field number, wire type, value): varint -> int, fixed64 -> 8 bytes, length-delimited -> bytes."""
    i, n = 0, len(buf)

    def varint():
        nonlocal i
        v = shift = 0
        while True:
            if i >= n:
                raise ValueError("truncated RuleTable message")
            b = buf[i]
            i += 1
            v |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return v

    while i < n:
        key = varint()
        num, wt = key >> 3, key & 7
        if wt == 0:
            yield num, wt, varint()
        elif wt == 2:
            ln = varint()
            if i + ln > n:
                raise ValueError("truncated RuleTable message")
            yield num, wt, bytes(buf[i:i + ln])
            i += ln
        elif wt == 1:
            yield num, wt, bytes(buf[i:i + 8])
            i += 8
        elif wt == 5:
            yield num, wt, bytes(buf[i:i + 4])
            i += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)


# ---- google.protobuf.Value -------------------------------------------------------------------------------------
def _enc_value(v) -> bytes:
    if v is None:
        return _varint(1 << 3) + b"\x00"
    if isinstance(v, bool):
        return _varint(4 << 3) + (b"\x01" if v else b"\x00")
    if isinstance(v, (int, float)):
        return _varint((2 << 3) | 1) + struct.pack("<d", float(v))
    if isinstance(v, str):
        return _ld(3, v.encode("utf-8"))
    if isinstance(v, dict):
        return _ld(5, b"".join(_ld(1, _ld(1, str(k).encode("utf-8")) + _ld(2, _enc_value(x))) for k, x in v.items()))
    if isinstance(v, (list, tuple)):
        return _ld(6, b"".join(_ld(1, _enc_value(x)) for x in v))
    raise TypeError("unsupported constant %r" % (v,))


def _dec_value(buf):
    out = None
    for num, wt, v in _fields(buf):
        if num == 1 and wt == 0:
            out = None
        elif num == 2 and wt == 1:
            out = struct.unpack("<d", v)[0]
            if out == int(out) and abs(out) < 2 ** 53:
                out = int(out)   # YAML integers come back as integers (constants print the same either way)
        elif num == 3 and wt == 2:
            out = v.decode("utf-8")
        elif num == 4 and wt == 0:
            out = bool(v)
        elif num == 5 and wt == 2:
            out = dict(_dec_map_entry(e, lambda b: b.decode("utf-8"), _dec_value) for n2, w2, e in _fields(v) if n2 == 1 and w2 == 2)
        elif num == 6 and wt == 2:
            out = [_dec_value(e) for n2, w2, e in _fields(v) if n2 == 1 and w2 == 2]
    return out


def _dec_map_entry(buf, key_fn, val_fn, default=None):
    k, v = key_fn(b"") if key_fn is not int else 0, default
    for num, wt, x in _fields(buf):
        if num == 1:
            k = x if wt == 0 else key_fn(x)
        elif num == 2 and wt == 2:
            v = val_fn(x)
    return k, v


# ---- Expr / Condition / Output / Params --------------------------------------------------------------------------
def _enc_expr(text: str) -> bytes:
    return _str(1, text)      # Expr.original; `checked` is the Go side's cache of its type-checked AST


def _dec_expr(buf) -> str:
    for num, wt, v in _fields(buf):
        if num == 1 and wt == 2:
            return v.decode("utf-8")
    return ""


def _enc_cond(c) -> bytes:
    if c[0] == "expr":
        return _ld(4, _enc_expr(c[1]))
    return _ld(_OPS[c[0]], b"".join(_ld(1, _enc_cond(x)) for x in c[1]))


def _dec_cond(buf):
    for num, wt, v in _fields(buf):
        if wt != 2:
            continue
        if num == 4:
            return ("expr", _dec_expr(v))
        if num in _OP_NAMES:
            return (_OP_NAMES[num], tuple(_dec_cond(x) for n2, w2, x in _fields(v) if n2 == 1 and w2 == 2))
    raise ValueError("empty runtimev1.Condition")


def _enc_output(o) -> bytes:
    when = b""
    if o.get("rule_activated"):
        when += _ld(1, _enc_expr(o["rule_activated"]))
    if o.get("condition_not_met"):
        when += _ld(2, _enc_expr(o["condition_not_met"]))
    return _ld(1, when)


def _dec_output(buf):
    out = {}
    for num, wt, v in _fields(buf):
        if num == 1 and wt == 2:
            for n2, w2, x in _fields(v):
                if w2 == 2 and n2 in (1, 2):
                    out["rule_activated" if n2 == 1 else "condition_not_met"] = _dec_expr(x)
    return out


def _enc_params(p) -> bytes:
    out = b"".join(_ld(1, _str(1, name) + _ld(2, _enc_expr(text))) for name, text in p["ordered_variables"])
    out += b"".join(_ld(2, _ld(1, str(k).encode("utf-8")) + _ld(2, _enc_value(v))) for k, v in p["constants"].items())
    return out


def _dec_params(buf):
    ov, consts = [], {}
    for num, wt, v in _fields(buf):
        if num == 1 and wt == 2:
            name, text = "", ""
            for n2, w2, x in _fields(v):
                if n2 == 1 and w2 == 2:
                    name = x.decode("utf-8")
                elif n2 == 2 and w2 == 2:
                    text = _dec_expr(x)
            ov.append((name, text))
        elif num == 2 and wt == 2:
            k, val = _dec_map_entry(v, lambda b: b.decode("utf-8"), _dec_value)
            consts[k] = val
    return {"ordered_variables": ov, "constants": consts}


_EK_FIELDS = (1, 2, 3, 4, 5, 6, 7, 8)   # EvaluationKeyTuple string fields in tuple order (runtime.proto:29-39)


def _enc_eval_key(t) -> bytes:
    return b"".join(_str(f, t[i]) for i, f in enumerate(_EK_FIELDS)) + _uint(9, t[8])


def _dec_eval_key(buf):
    vals = [""] * 8 + [0]
    for num, wt, v in _fields(buf):
        if 1 <= num <= 8 and wt == 2:
            vals[num - 1] = v.decode("utf-8")
        elif num == 9 and wt == 0:
            vals[8] = v
    return tuple(vals)


# ---- RuleRow -------------------------------------------------------------------------------------------------------
def _enc_row(r) -> bytes:
    out = _str(1, r["origin_fqn"]) + _str(2, r["resource"]) + _str(3, r["role"])
    if r["allow_actions"] is not None:
        out += _ld(15, b"".join(_ld(1, _ld(1, a.encode("utf-8")) + _ld(2, b"")) for a in r["allow_actions"]))
    elif r["action"] is not None:
        out += _ld(4, r["action"].encode("utf-8"))     # oneof member: present even when empty
    if r["condition"] is not None:
        out += _ld(5, _enc_cond(r["condition"]))
    if r["derived_role_condition"] is not None:
        out += _ld(6, _enc_cond(r["derived_role_condition"]))
    out += _uint(7, _EFFECT_NUM.get(r["effect"], 0)) + _str(8, r["scope"]) + _uint(9, r["scope_permissions"])
    out += _str(10, r["version"]) + _str(11, r["origin_derived_role"])
    if r["emit_output"]:
        out += _ld(12, _enc_output(r["emit_output"]))
    out += _str(13, r["name"]) + _str(14, r["principal"])
    if r["params"] is not None:
        out += _ld(16, _enc_params(r["params"]))
    if r["derived_role_params"] is not None:
        out += _ld(17, _enc_params(r["derived_role_params"]))
    out += _uint(19, _KIND_NUM[r["policy_kind"]]) + _uint(20, 1 if r["from_role_policy"] else 0)
    if r["evaluation_key"] is not None:
        out += _ld(21, _enc_eval_key(r["evaluation_key"]))
    return out


def _dec_row(buf):
    r = {"origin_fqn": "", "resource": "", "role": "", "action": None, "allow_actions": None, "condition": None,
         "derived_role_condition": None, "effect": None, "scope": "", "scope_permissions": 0, "version": "",
         "origin_derived_role": "", "emit_output": None, "name": "", "principal": "", "params": None,
         "derived_role_params": None, "evaluation_key": None, "policy_kind": KIND_RESOURCE, "from_role_policy": False}
    legacy_key = None
    for num, wt, v in _fields(buf):
        if wt == 2:
            if num == 1:
                r["origin_fqn"] = v.decode("utf-8")
            elif num == 2:
                r["resource"] = v.decode("utf-8")
            elif num == 3:
                r["role"] = v.decode("utf-8")
            elif num == 4:
                r["action"] = v.decode("utf-8")
            elif num == 15:
                r["allow_actions"] = [_dec_map_entry(e, lambda b: b.decode("utf-8"), lambda b: None)[0]
                                      for n2, w2, e in _fields(v) if n2 == 1 and w2 == 2]
            elif num == 5:
                r["condition"] = _dec_cond(v)
            elif num == 6:
                r["derived_role_condition"] = _dec_cond(v)
            elif num == 8:
                r["scope"] = v.decode("utf-8")
            elif num == 10:
                r["version"] = v.decode("utf-8")
            elif num == 11:
                r["origin_derived_role"] = v.decode("utf-8")
            elif num == 12:
                r["emit_output"] = _dec_output(v) or None
            elif num == 13:
                r["name"] = v.decode("utf-8")
            elif num == 14:
                r["principal"] = v.decode("utf-8")
            elif num == 16:
                r["params"] = _dec_params(v)
            elif num == 17:
                r["derived_role_params"] = _dec_params(v)
            elif num == 18:
                legacy_key = v.decode("utf-8")
            elif num == 21:
                r["evaluation_key"] = _dec_eval_key(v)
        elif wt == 0:
            if num == 7:
                r["effect"] = _EFFECT_NAME.get(v)
            elif num == 9:
                r["scope_permissions"] = v
            elif num == 19:
                r["policy_kind"] = _KIND_NAME.get(v, KIND_RESOURCE)
            elif num == 20:
                r["from_role_policy"] = bool(v)
    if r["evaluation_key"] is None and legacy_key is not None:
        # tables serialized before the tuple existed: the string key alone, as the rule name of an otherwise empty tuple
        # (index/core.go:134-137 makeEvaluationKeyTuple's fallback)
        r["evaluation_key"] = ("",) * 7 + (legacy_key, 0)
    return r


# ---- side tables -----------------------------------------------------------------------------------------------------
def _enc_derived_role(dr) -> bytes:
    out = _str(1, dr["name"])
    out += b"".join(_ld(2, _ld(1, p.encode("utf-8")) + _ld(2, b"")) for p in dr["parent_roles"])
    if dr["condition"] is not None:
        out += _ld(4, _enc_cond(dr["condition"]))
    out += b"".join(_ld(5, _str(1, n) + _ld(2, _enc_expr(t))) for n, t in dr["ordered_variables"])
    out += b"".join(_ld(6, _ld(1, str(k).encode("utf-8")) + _ld(2, _enc_value(v))) for k, v in dr["constants"].items())
    return out + _str(7, dr["origin_fqn"])


def _dec_derived_role(buf):
    dr = {"name": "", "parent_roles": [], "origin_fqn": "", "condition": None, "constants": {}, "ordered_variables": []}
    for num, wt, v in _fields(buf):
        if wt != 2:
            continue
        if num == 1:
            dr["name"] = v.decode("utf-8")
        elif num == 2:
            dr["parent_roles"].append(_dec_map_entry(v, lambda b: b.decode("utf-8"), lambda b: None)[0])
        elif num == 4:
            dr["condition"] = _dec_cond(v)
        elif num == 5:
            name, text = "", ""
            for n2, w2, x in _fields(v):
                if n2 == 1 and w2 == 2:
                    name = x.decode("utf-8")
                elif n2 == 2 and w2 == 2:
                    text = _dec_expr(x)
            dr["ordered_variables"].append((name, text))
        elif num == 6:
            k, val = _dec_map_entry(v, lambda b: b.decode("utf-8"), _dec_value)
            dr["constants"][k] = val
        elif num == 7:
            dr["origin_fqn"] = v.decode("utf-8")
    return dr


_META_NAME_FIELD = {"resource": 2, "role": 3, "principal": 7}   # RuleTableMetadata oneof name (runtime.proto:107-117)
_META_KIND = {v: k for k, v in _META_NAME_FIELD.items()}


def encode_rule_table(rt: dict) -> bytes:
    """The rule-table dict as serialized ``runtimev1.RuleTable`` (rules 1, meta 3, scope_parent_roles 4,
    policy_derived_roles 5).  Schemas (2, 6), the bundle manifest (7) and the compiler version (8) are not part of
    the decision path and are left out."""
    out = bytearray()
    for r in rt["rules"]:
        out += _ld(1, _enc_row(r))
    for fqn, m in rt["meta"].items():
        body = _str(1, m["fqn"]) + _str(_META_NAME_FIELD[m["kind"]], m["name"]) + _str(4, m["version"])
        out += _ld(3, _varint(1 << 3) + _varint(module_id(fqn)) + _ld(2, body))
    for scope, roles in rt["scope_parent_roles"].items():
        inner = b"".join(_ld(1, _ld(1, role.encode("utf-8")) + _ld(2, b"".join(_str(1, p) for p in parents)))
                         for role, parents in roles.items())
        out += _ld(4, _ld(1, scope.encode("utf-8")) + _ld(2, inner))
    for fqn, drs in rt["policy_derived_roles"].items():
        inner = b"".join(_ld(1, _ld(1, name.encode("utf-8")) + _ld(2, _enc_derived_role(dr))) for name, dr in drs.items())
        out += _ld(5, _varint(1 << 3) + _varint(module_id(fqn)) + _ld(2, inner))
    return bytes(out)


def decode_rule_table(buf: bytes) -> dict:
    """Serialized ``runtimev1.RuleTable`` -> the dict ``build_rule_table`` returns (rows numbered in message order)."""
    rt = {"rules": [], "meta": {}, "scope_parent_roles": {}, "policy_derived_roles": {}}
    metas, pdr = {}, []
    for num, wt, v in _fields(buf):
        if wt != 2:
            continue
        if num == 1:
            rt["rules"].append(_dec_row(v))
        elif num == 3:
            mid, body = _dec_map_entry(v, int, lambda b: b)
            m = {"fqn": "", "kind": "resource", "name": "", "version": ""}
            for n2, w2, x in _fields(body or b""):
                if w2 != 2:
                    continue
                if n2 == 1:
                    m["fqn"] = x.decode("utf-8")
                elif n2 in _META_KIND:
                    m["kind"], m["name"] = _META_KIND[n2], x.decode("utf-8")
                elif n2 == 4:
                    m["version"] = x.decode("utf-8")
            metas[mid] = m
        elif num == 4:
            scope, inner = _dec_map_entry(v, lambda b: b.decode("utf-8"), lambda b: b)
            roles = {}
            for n2, w2, e in _fields(inner or b""):
                if n2 == 1 and w2 == 2:
                    role, parents = _dec_map_entry(e, lambda b: b.decode("utf-8"),
                                                   lambda b: [x.decode("utf-8") for n3, w3, x in _fields(b) if n3 == 1 and w3 == 2], [])
                    roles[role] = parents
            rt["scope_parent_roles"][scope] = roles
        elif num == 5:
            mid, inner = _dec_map_entry(v, int, lambda b: b)
            drs = {}
            for n2, w2, e in _fields(inner or b""):
                if n2 == 1 and w2 == 2:
                    name, dr = _dec_map_entry(e, lambda b: b.decode("utf-8"), _dec_derived_role)
                    drs[name] = dr
            pdr.append((mid, drs))
    for m in metas.values():
        rt["meta"][m["fqn"]] = m
    for mid, drs in pdr:
        if mid not in metas:
            raise ValueError("policy_derived_roles refers to module %d, which has no metadata entry" % mid)
        rt["policy_derived_roles"][metas[mid]["fqn"]] = drs
    for i, row in enumerate(rt["rules"]):
        row["id"] = i
    return finish_rule_table(rt)
