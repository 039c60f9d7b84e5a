"""Columnar request batches: the pre-flattened ingest path of the boundary.

``flatten.Flattener`` walks JSON-shaped CheckInputs one by one (what a drop-in behind the
gRPC handler sees).  Bulk callers - and the benchmarks, which need millions of tuples -
already hold their requests as columns (ids, role lists, attribute columns).  This module
describes such a batch (``ColumnarRequests``) and turns it into the same ``cbh_batch`` SoA
with vectorised numpy operations; ``to_inputs`` materialises the equivalent CheckInput
dicts so both routes can be compared (tests/test_columnar.py) and the CPU oracle can be
run on samples.
"""
from __future__ import annotations

import numpy as np

from . import namer
from .flatten import (HEAP_BATCH, MAX_ACTIONS_PER_REQUEST, RQ_ACT_CNT, RQ_ACT_OFF, RQ_NFIELDS, RQ_KIND,
                      RQ_P_SCOPE, RQ_P_VERSION, RQ_PRINCIPAL_ID, sort_batch_by_route,
                      RQ_R_SCOPE, RQ_R_VERSION, RQ_ROLE_CNT, RQ_ROLE_OFF, RQ_S_KIND, RQ_S_P_SCOPE,
                      RQ_S_P_VERSION, RQ_S_R_SCOPE, RQ_S_R_VERSION, RQ_S_RESOURCE_ID, SF_ACTION,
                      SF_KIND, SF_ROLE, T_ABSENT, T_BOOL, T_DOUBLE, T_ERR, T_LIST, T_MAP, T_NULL,
                      T_STRING, Batch, Flattener, _f64_bits)


class Vocab:
    """A string-valued column: ``values[idx[i]]`` is the string of request i."""

    def __init__(self, values, idx):
        self.values = list(values)
        self.idx = np.asarray(idx, dtype=np.int64)

    def at(self, i):
        return self.values[self.idx[i]]


class Ragged:
    """A list-of-strings column (roles, actions): request i holds
    ``values[flat[off[i]:off[i+1]]]``."""

    def __init__(self, values, off, flat):
        self.values = list(values)
        self.off = np.asarray(off, dtype=np.int64)
        self.flat = np.asarray(flat, dtype=np.int64)

    def at(self, i):
        return [self.values[j] for j in self.flat[self.off[i]:self.off[i + 1]]]


class Attr:
    """One attribute column.  kind: 'str' (vocab indices), 'num' (float64), 'bool',
    'json' (indices into a vocabulary of arbitrary JSON values).  ``present`` masks
    requests that carry the attribute at all."""

    def __init__(self, kind, data, present=None, values=None):
        self.kind = kind
        self.data = np.asarray(data)
        self.values = list(values) if values is not None else None
        self.present = None if present is None else np.asarray(present, dtype=bool)

    def at(self, i):
        if self.present is not None and not self.present[i]:
            return None, False
        if self.kind == "num":
            return float(self.data[i]), True
        if self.kind == "bool":
            return bool(self.data[i]), True
        return self.values[int(self.data[i])], True


class ColumnarRequests:
    def __init__(self, n, principal_id: Vocab, roles: Ragged, resource_kind: Vocab, resource_id: Vocab,
                 actions: Ragged, p_attr=None, r_attr=None, principal_scope: Vocab = None,
                 resource_scope: Vocab = None, principal_version: Vocab = None, resource_version: Vocab = None):
        self.n = n
        self.principal_id, self.roles = principal_id, roles
        self.resource_kind, self.resource_id, self.actions = resource_kind, resource_id, actions
        self.p_attr, self.r_attr = dict(p_attr or {}), dict(r_attr or {})
        self.principal_scope, self.resource_scope = principal_scope, resource_scope
        self.principal_version, self.resource_version = principal_version, resource_version

    def head(self, n):
        """The first ``n`` requests as a new ColumnarRequests."""
        n = min(n, self.n)

        def voc(v):
            return None if v is None else Vocab(v.values, v.idx[:n])

        def rag(r):
            return Ragged(r.values, r.off[:n + 1], r.flat[:r.off[n]])

        def att(d):
            return {k: Attr(a.kind, a.data[:n], None if a.present is None else a.present[:n], a.values)
                    for k, a in d.items()}

        return ColumnarRequests(n, voc(self.principal_id), rag(self.roles), voc(self.resource_kind),
                                voc(self.resource_id), rag(self.actions), att(self.p_attr), att(self.r_attr),
                                voc(self.principal_scope), voc(self.resource_scope),
                                voc(self.principal_version), voc(self.resource_version))

    # ---- dict route ---------------------------------------------------------------------
    def to_inputs(self, start=0, stop=None):
        stop = self.n if stop is None else min(stop, self.n)
        out = []
        for i in range(start, stop):
            p = {"id": self.principal_id.at(i), "roles": self.roles.at(i), "attr": {}}
            r = {"kind": self.resource_kind.at(i), "id": self.resource_id.at(i), "attr": {}}
            for name, col in self.p_attr.items():
                v, ok = col.at(i)
                if ok:
                    p["attr"][name] = v
            for name, col in self.r_attr.items():
                v, ok = col.at(i)
                if ok:
                    r["attr"][name] = v
            if self.principal_scope is not None:
                p["scope"] = self.principal_scope.at(i)
            if self.resource_scope is not None:
                r["scope"] = self.resource_scope.at(i)
            if self.principal_version is not None:
                p["policyVersion"] = self.principal_version.at(i)
            if self.resource_version is not None:
                r["policyVersion"] = self.resource_version.at(i)
            out.append({"requestId": "q%d" % i, "principal": p, "resource": r, "actions": self.actions.at(i)})
        return out

    # ---- SoA route ----------------------------------------------------------------------
    def to_batch(self, fl: Flattener, default_policy_version="default", default_scope="", sort=True) -> Batch:  # noqa: C901
        lt, K, n = fl.lt, fl.K, self.n
        table_ids = lt.string_ids
        local, local_strings, local_flags = {}, [], []

        def sid(s, flag=0):
            i = table_ids.get(s)
            if i is not None:
                return i
            j = local.get(s)
            if j is None:
                j = len(local_strings)
                local[s] = j
                local_strings.append(s)
                local_flags.append(flag)
            elif flag:
                local_flags[j] |= flag
            return K + j

        heap_tag, heap_val = [], []

        def enc(v):
            if v is None:
                return T_NULL, 0
            if isinstance(v, bool):
                return T_BOOL, int(v)
            if isinstance(v, (int, float)):
                return T_DOUBLE, _f64_bits(v)
            if isinstance(v, str):
                return T_STRING, sid(v)
            if isinstance(v, (list, tuple)):
                vals = [enc(x) for x in v]
                off = len(heap_tag)
                for t, pv in vals:
                    heap_tag.append(t)
                    heap_val.append(pv)
                return T_LIST, (HEAP_BATCH << 62) | (off << 32) | len(vals)
            if isinstance(v, dict):
                ents = [((T_STRING, sid(str(k))), enc(x)) for k, x in v.items()]
                off = len(heap_tag)
                for (kt, kp), (vt, vp) in ents:
                    heap_tag.extend((kt, vt))
                    heap_val.extend((kp, vp))
                return T_MAP, (HEAP_BATCH << 62) | (off << 32) | len(ents)
            raise TypeError("unsupported attribute value %r" % (v,))

        def vocab_ids(v: Vocab, fn=lambda s: s, flag=0):
            m = np.array([sid(fn(s), flag) for s in v.values], dtype=np.uint32)
            return m[v.idx]

        def const_col(s):
            return np.full(n, sid(s), dtype=np.uint32)

        req = np.zeros((RQ_NFIELDS, n), dtype=np.uint32)
        req[RQ_PRINCIPAL_ID] = vocab_ids(self.principal_id)
        req[RQ_KIND] = vocab_ids(self.resource_kind, namer.sanitize, SF_KIND)
        req[RQ_S_KIND] = vocab_ids(self.resource_kind)
        req[RQ_S_RESOURCE_ID] = vocab_ids(self.resource_id)

        def scope_cols(v: Vocab, f_word, f_str):
            if v is None:
                eff = namer.scope_value(default_scope)
                req[f_word] = fl.scope_word(eff)
                req[f_str] = sid("")
                return
            words = np.array([fl.scope_word(namer.scope_value(s if s != "" else default_scope)) for s in v.values],
                             dtype=np.uint32)
            req[f_word] = words[v.idx]
            req[f_str] = vocab_ids(v, namer.scope_value)

        scope_cols(self.principal_scope, RQ_P_SCOPE, RQ_S_P_SCOPE)
        scope_cols(self.resource_scope, RQ_R_SCOPE, RQ_S_R_SCOPE)

        def version_cols(v: Vocab, f_eff, f_raw):
            if v is None:
                req[f_eff] = sid(default_policy_version)
                req[f_raw] = sid("")
                return
            req[f_eff] = vocab_ids(v, lambda s: s or default_policy_version)
            req[f_raw] = vocab_ids(v)

        version_cols(self.principal_version, RQ_P_VERSION, RQ_S_P_VERSION)
        version_cols(self.resource_version, RQ_R_VERSION, RQ_S_R_VERSION)

        role_ids = np.array([sid(s, SF_ROLE) for s in self.roles.values], dtype=np.uint32)
        roles = role_ids[self.roles.flat]
        req[RQ_ROLE_OFF] = self.roles.off[:-1].astype(np.uint32)
        req[RQ_ROLE_CNT] = np.diff(self.roles.off).astype(np.uint32)

        act_ids = np.array([sid(s, SF_ACTION) for s in self.actions.values], dtype=np.uint32)
        counts = np.diff(self.actions.off)
        if counts.size and counts.max() > MAX_ACTIONS_PER_REQUEST:
            raise ValueError("the columnar route carries at most %d actions per request" % MAX_ACTIONS_PER_REQUEST)
        req[RQ_ACT_OFF] = self.actions.off[:-1].astype(np.uint32)
        req[RQ_ACT_CNT] = counts.astype(np.uint32)
        tuple_req = np.repeat(np.arange(n, dtype=np.uint32), counts)
        tuple_action = act_ids[self.actions.flat]

        ncol = len(lt.columns)
        col_tag = np.full((ncol, n), T_ABSENT, dtype=np.uint8)
        col_val = np.zeros((ncol, n), dtype=np.uint64)
        for ci, (root, keys) in enumerate(lt.columns):
            attrs = self.p_attr if root == "P" else (self.r_attr if root == "R" else {})
            if not keys:
                raise NotImplementedError("whole-attribute-map columns are not supported by the columnar route")
            col = attrs.get(keys[0])
            if col is None:
                if len(keys) > 1:
                    col_tag[ci] = T_ERR  # intermediate key missing
                continue
            if len(keys) == 1 and col.kind == "num":
                tags = np.full(n, T_DOUBLE, dtype=np.uint8)
                vals = col.data.astype(np.float64).view(np.uint64)
            elif len(keys) == 1 and col.kind == "bool":
                tags = np.full(n, T_BOOL, dtype=np.uint8)
                vals = col.data.astype(np.uint64)
            else:
                # per-vocabulary-entry walk (strings, json values, nested paths)
                vt = np.zeros(len(col.values), dtype=np.uint8)
                vv = np.zeros(len(col.values), dtype=np.uint64)
                for k, v in enumerate(col.values):
                    cur, tag = v, None
                    for ki, key in enumerate(keys[1:]):
                        if not isinstance(cur, dict):
                            tag = T_ERR
                            break
                        if key not in cur:
                            tag = T_ABSENT if ki == len(keys) - 2 else T_ERR
                            break
                        cur = cur[key]
                    if tag is None:
                        tag, val = enc(cur)
                        vv[k] = val
                    vt[k] = tag
                tags, vals = vt[col.data.astype(np.int64)], vv[col.data.astype(np.int64)]
            if col.present is not None:
                miss = T_ABSENT if len(keys) == 1 else T_ERR
                tags = np.where(col.present, tags, miss).astype(np.uint8)
                vals = np.where(col.present, vals, 0).astype(np.uint64)
            col_tag[ci], col_val[ci] = tags, vals

        b = Batch()
        b.n_requests, b.n_tuples = n, int(tuple_req.size)
        b.req_u32 = np.ascontiguousarray(req)
        b.roles = np.ascontiguousarray(roles, dtype=np.uint32)
        b.tuple_req, b.tuple_action = tuple_req, np.ascontiguousarray(tuple_action, dtype=np.uint32)
        b.col_tag, b.col_val = np.ascontiguousarray(col_tag), np.ascontiguousarray(col_val)
        b.heap_tag = np.asarray(heap_tag, dtype=np.uint8)
        b.heap_val = np.array([int(x) for x in heap_val], dtype=np.uint64)
        encs = [s.encode("utf-8") for s in local_strings]
        off = np.zeros(len(encs) + 1, dtype=np.uint32)
        if encs:
            off[1:] = np.cumsum([len(x) for x in encs])
        b.str_off = off
        b.str_bytes = np.frombuffer(b"".join(encs), dtype=np.uint8).copy()
        b.str_flags = np.asarray(local_flags, dtype=np.uint8)
        b.n_strings = len(local_strings)
        b.actions_per_request = None  # use actions.at(i) when decoding
        return sort_batch_by_route(b) if sort else b
