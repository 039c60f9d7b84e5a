"""CheckInput batch -> SoA device batch (host side of the boundary).

Mirrors what ``check()`` derives from a ``CheckInput`` before it touches the rule table
(``internal/ruletable/check.go:101-117, 155, 164, 536-554``): effective scope / version,
sanitised kind, the request view CEL programs read - but for a whole batch at once and as
integer ids.  Strings are interned against the table's own string pool so that on the
device string equality is id equality; strings the table has never seen get batch-local
ids and are shipped with the batch (their glob match bits are resolved on the device).

The device evaluates one request per lane and walks the rule table once per group of lanes
that share a routing key (resource kind, policy version, scope), so requests are ordered by
that key here (``sort_batch_by_route``); results are mapped back to input order by
``Batch.tuple_perm`` / ``Batch.req_perm``.

This Python flattener is the reference implementation of the ``cbh_batch`` contract; the
per-batch cost is host-side and outside the GPU timed region (see DESIGN.md - the C++
wire-format flattener is SURVEY.md §8(f) rank 1).
"""
from __future__ import annotations

import struct

import numpy as np

from . import namer
from .lower.blob import LoweredTable

RQ_NFIELDS = 16
(RQ_PRINCIPAL_ID, RQ_P_SCOPE, RQ_P_VERSION, RQ_KIND, RQ_R_SCOPE, RQ_R_VERSION, RQ_ROLE_OFF, RQ_ROLE_CNT,
 RQ_ACT_OFF, RQ_ACT_CNT,
 RQ_S_RESOURCE_ID, RQ_S_KIND, RQ_S_P_SCOPE, RQ_S_R_SCOPE, RQ_S_P_VERSION, RQ_S_R_VERSION) = range(16)
SCOPE_EXACT = 0x80000000
MAX_ACTIONS_PER_REQUEST = 64

T_NULL, T_BOOL, T_INT, T_UINT, T_DOUBLE, T_STRING, T_LIST, T_MAP = range(8)
T_ABSENT, T_ERR = 0xF0, 0xFF
HEAP_BATCH = 1
SF_ACTION, SF_ROLE, SF_KIND = 1, 2, 4


def _f64_bits(x: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


class Batch:
    """Host image of a ``cbh_batch`` plus what is needed to decode results."""

    def __init__(self):
        self.n_requests = 0      # device (virtual) requests
        self.n_tuples = 0
        self.req_u32 = None
        self.roles = None
        self.tuple_req = None
        self.tuple_action = None
        self.col_tag = None
        self.col_val = None
        self.heap_tag = None
        self.heap_val = None
        self.str_off = None
        self.str_bytes = None
        self.str_flags = None
        self.n_strings = 0
        self.actions_per_request = []  # [[action, ...]] per INPUT, input order
        self.tuple_perm = None   # device tuple j holds input-order tuple tuple_perm[j] (None = identity)
        self.req_perm = None     # device request i is pre-sort request req_perm[i]
        self.vreq_input = None   # pre-sort (virtual) request -> index of the CheckInput it came from


def sort_batch_by_route(b: Batch) -> Batch:
    """Order requests by (kind, resource version, resource scope), then by role list, so that the
    lanes of a wave share their policy buckets AND walk them the same number of times (the kernel
    visits a bucket once per role index any lane still needs, check.go:208); regroups the tuple
    arrays accordingly (in place)."""
    n = b.n_requests
    if n < 2:
        return b
    req = b.req_u32
    cnt = req[RQ_ROLE_CNT].astype(np.int64)
    sig = np.zeros(n, dtype=np.uint64)
    if b.roles is not None and len(b.roles) and cnt.any():
        off = req[RQ_ROLE_OFF].astype(np.int64)
        start = np.cumsum(cnt) - cnt
        pos = np.arange(int(cnt.sum()), dtype=np.int64) - np.repeat(start, cnt)   # index of a role within its list
        with np.errstate(over="ignore"):
            w = (np.asarray(b.roles).astype(np.uint64)[np.repeat(off, cnt) + pos] + np.uint64(1)) \
                * (pos.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xC2B2AE3D27D4EB4F))
            csum = np.concatenate([np.zeros(1, dtype=np.uint64), np.cumsum(w, dtype=np.uint64)])
            sig = csum[start + cnt] - csum[start]                                 # order-sensitive list signature
    order = np.lexsort((sig, cnt, req[RQ_R_SCOPE], req[RQ_R_VERSION], req[RQ_KIND]))   # stable
    if np.array_equal(order, np.arange(n)):
        return b
    return permute_requests(b, order)


def permute_requests(b: Batch, order) -> Batch:
    """Device request i becomes the current request order[i]; tuple arrays are regrouped and the
    permutations recorded so results can be returned in input order (in place)."""
    n = b.n_requests
    req = b.req_u32
    order = np.asarray(order, dtype=np.int64)
    counts = req[RQ_ACT_CNT][order].astype(np.int64)
    starts = req[RQ_ACT_OFF][order].astype(np.int64)
    new_off = np.cumsum(counts) - counts
    total = int(counts.sum())
    src = np.repeat(starts - new_off, counts) + np.arange(total, dtype=np.int64)   # device tuple j <- old tuple src[j]
    b.req_u32 = np.ascontiguousarray(req[:, order])
    b.req_u32[RQ_ACT_OFF] = new_off.astype(np.uint32)
    b.col_tag = np.ascontiguousarray(b.col_tag[:, order])
    b.col_val = np.ascontiguousarray(b.col_val[:, order])
    b.tuple_action = np.ascontiguousarray(b.tuple_action[src])
    b.tuple_req = np.repeat(np.arange(n, dtype=np.uint32), counts)
    b.tuple_perm = src if b.tuple_perm is None else b.tuple_perm[src]
    b.req_perm = order if b.req_perm is None else b.req_perm[order]
    return b


class Flattener:
    def __init__(self, lt: LoweredTable):
        self.lt = lt
        self.K = len(lt.strings)
        self._scope_cache = {}

    def scope_word(self, scope: str) -> int:
        """(index of the nearest table-known ancestor-or-self) | EXACT bit."""
        w = self._scope_cache.get(scope)
        if w is None:
            si = self.lt.scope_index.get(scope)
            if si is not None:
                w = si | SCOPE_EXACT
            else:
                w = 0
                for p in namer.scope_parents(scope):
                    si = self.lt.scope_index.get(p)
                    if si is not None:
                        w = si
                        break
            self._scope_cache[scope] = w
        return w

    def flatten(self, inputs, default_policy_version="default", default_scope="", sort=True, globals_=None) -> Batch:
        lt, K = self.lt, self.K
        table_ids = lt.string_ids
        local = {}
        local_strings = []
        local_flags = []

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
                return T_DOUBLE, _f64_bits(v)          # structpb: every number is a double
            if isinstance(v, str):
                return T_STRING, sid(v)
            if isinstance(v, (list, tuple)):
                vals = [enc(x) for x in v]
                off = len(heap_tag)
                for t, p in vals:
                    heap_tag.append(t)
                    heap_val.append(p)
                return T_LIST, (HEAP_BATCH << 62) | (off << 32) | len(vals)
            if isinstance(v, dict):
                ents = [((T_STRING, sid(str(k))), enc(x)) for k, x in v.items()]
                off = len(heap_tag)
                for (kt, kp), (vt, vp) in ents:
                    heap_tag.extend((kt, vt))
                    heap_val.extend((kp, vp))
                return T_MAP, (HEAP_BATCH << 62) | (off << 32) | len(ents)
            raise TypeError("unsupported attribute value %r" % (v,))

        # a CheckInput with more than 64 actions becomes several device requests
        chunks = []
        for i, inp in enumerate(inputs):
            acts = list(inp.get("actions") or [])
            if not acts:
                chunks.append((i, acts))
            for s in range(0, len(acts), MAX_ACTIONS_PER_REQUEST):
                chunks.append((i, acts[s:s + MAX_ACTIONS_PER_REQUEST]))
        n = len(chunks)
        ncol = len(lt.columns)
        req = np.zeros((RQ_NFIELDS, n), dtype=np.uint32)
        col_tag = np.full((ncol, n), T_ABSENT, dtype=np.uint8)
        col_val = np.zeros((ncol, n), dtype=np.uint64)
        roles, t_req, t_act = [], [], []
        b = Batch()
        b.actions_per_request = [list(inp.get("actions") or []) for inp in inputs]
        b.vreq_input = np.array([i for i, _ in chunks], dtype=np.int64)
        b.vreq_actions = [acts for _, acts in chunks]   # per device request (before any routing sort): its action names
        for r, (i_in, acts) in enumerate(chunks):
            inp = inputs[i_in]
            p, res = inp["principal"], inp["resource"]
            aux = inp.get("auxData") or {}
            p_scope_raw = p.get("scope", "") or ""
            r_scope_raw = res.get("scope", "") or ""
            p_scope = namer.scope_value(p_scope_raw if p_scope_raw != "" else default_scope)
            r_scope = namer.scope_value(r_scope_raw if r_scope_raw != "" else default_scope)
            p_ver = p.get("policyVersion", "") or default_policy_version
            r_ver = res.get("policyVersion", "") or default_policy_version
            req[RQ_PRINCIPAL_ID, r] = sid(p.get("id", ""))
            req[RQ_P_SCOPE, r] = self.scope_word(p_scope)
            req[RQ_P_VERSION, r] = sid(p_ver)
            req[RQ_KIND, r] = sid(namer.sanitize(res.get("kind", "")), SF_KIND)
            req[RQ_R_SCOPE, r] = self.scope_word(r_scope)
            req[RQ_R_VERSION, r] = sid(r_ver)
            prs = list(p.get("roles") or [])
            req[RQ_ROLE_OFF, r] = len(roles)
            req[RQ_ROLE_CNT, r] = len(prs)
            roles.extend(sid(x, SF_ROLE) for x in prs)
            req[RQ_S_RESOURCE_ID, r] = sid(res.get("id", ""))
            req[RQ_S_KIND, r] = sid(res.get("kind", ""))
            req[RQ_S_P_SCOPE, r] = sid(namer.scope_value(p_scope_raw))
            req[RQ_S_R_SCOPE, r] = sid(namer.scope_value(r_scope_raw))
            req[RQ_S_P_VERSION, r] = sid(p.get("policyVersion", "") or "")
            req[RQ_S_R_VERSION, r] = sid(res.get("policyVersion", "") or "")
            roots = {"P": p.get("attr") or {}, "R": res.get("attr") or {}, "J": aux.get("jwt") or {},
                     # name -> {"claims": {...}}; a JWT without claims has an empty map (as the protobuf message does)
                     "S": {k: {"claims": (j or {}).get("claims") or {}} for k, j in (aux.get("jwts") or {}).items()},
                     "G": globals_ or {}}   # the CALL's globals (a table lowered with per_call_globals reads them as columns)
            for ci, (root, keys) in enumerate(lt.columns):
                cur = roots[root]
                tag = None
                for ki, key in enumerate(keys):
                    if not isinstance(cur, dict):
                        tag = T_ERR
                        break
                    if key not in cur:
                        tag = T_ABSENT if ki == len(keys) - 1 else T_ERR
                        break
                    cur = cur[key]
                if tag is None:
                    tag, val = enc(cur)
                    col_val[ci, r] = val
                col_tag[ci, r] = tag
            req[RQ_ACT_OFF, r] = len(t_act)
            req[RQ_ACT_CNT, r] = len(acts)
            for a in acts:
                t_req.append(r)
                t_act.append(sid(a, SF_ACTION))

        b.n_requests = n
        b.n_tuples = len(t_req)
        b.req_u32 = np.ascontiguousarray(req)
        b.roles = np.asarray(roles, dtype=np.uint32)
        b.tuple_req = np.asarray(t_req, dtype=np.uint32)
        b.tuple_action = np.asarray(t_act, dtype=np.uint32)
        b.col_tag = np.ascontiguousarray(col_tag)
        b.col_val = np.ascontiguousarray(col_val)
        b.heap_tag = np.asarray(heap_tag, dtype=np.uint8)
        b.heap_val = np.array([int(x) for x in heap_val], dtype=np.uint64)
        enc_strings = [s.encode("utf-8") for s in local_strings]
        off = np.zeros(len(enc_strings) + 1, dtype=np.uint32)
        if enc_strings:
            off[1:] = np.cumsum([len(x) for x in enc_strings])
        b.str_off = off
        b.str_bytes = np.frombuffer(b"".join(enc_strings), dtype=np.uint8).copy()
        b.str_flags = np.asarray(local_flags, dtype=np.uint8)
        b.n_strings = len(local_strings)
        return sort_batch_by_route(b) if sort else b
