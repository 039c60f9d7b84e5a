"""TEST INFRASTRUCTURE: the device flattener (cerbos_amd/csrc/cbh_wire.h) against the host flattener
(libcerbos_ingest.so cbi_flatten_pb, sort = 0) - the two must describe the same requests.  Ids of batch-local strings
differ by construction (first-appearance order on the host, dictionary slots on the device) and the device allocates
nested values in a different heap order, so the comparison is by MEANING: every id resolved to its bytes, every container
decoded recursively."""
import ctypes as C

import numpy as np

import hostsim_api
from cerbos_amd.flatten import Batch

RQ_STRING_FIELDS = (0, 2, 3, 5)            # principal id, principal version, kind, resource version
RQ_RAW_STRING_FIELDS = (10, 11, 12, 13, 14, 15)
RQ_PLAIN_FIELDS = (1, 4, 7, 9)             # scope words, role count, action count
T_STRING, T_LIST, T_MAP = 5, 6, 7
MF_READS_REQUEST_STRINGS = 64


class HsWire(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n", "n_tuples", "n_roles", "n_columns", "dict_slots", "heap_len", "K", "fill_runs")] + \
               [(n, C.c_void_p) for n in ("req_u32", "roles", "tuple_action", "col_tag", "col_val", "heap_tag", "heap_val", "dict",
                                          "dict_flags", "in_span", "act_span", "msg", "status")] + \
               [("stats", C.c_uint32 * 20)] + \
               [(n, C.c_void_p) for n in ("req_grouped", "col_tag_grouped", "col_val_grouped", "inv")] + [("n_routes", C.c_uint32), ("pad", C.c_uint32)]


STAT_NAMES = ("n_tuples", "n_roles", "max_actions", "max_roles", "wide_lo", "wide_hi", "first_bad", "n_host", "heap_used", "flags",
              "sid_empty", "sid_dver", "dscope_word", "sid_claims")


class WireBatch:
    """What the three kernels left behind, as numpy copies."""


def _arr(ptr, ctype, dtype, n):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True)


def sim_flatten(lt, data, offsets, default_version="default", default_scope="", dict_slots=0, heap=0, globals_pb=b""):
    """The device flattener on the host simulator.  Returns (rc, WireBatch): rc 1 = the table's inputs are the host's."""
    lib = hostsim_api.lib()
    lib.hostsim_wire_flatten.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t,
                                         C.c_uint32, C.c_uint32, C.POINTER(HsWire)]
    lib.hostsim_wire_flatten.restype = C.c_int
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out = HsWire()
    buf = C.create_string_buffer(lt.blob, len(lt.blob))
    pad = np.concatenate([data, np.zeros(8, np.uint8)])
    rc = lib.hostsim_wire_flatten(C.cast(buf, C.c_void_p), len(lt.blob), pad.ctypes.data, offsets.ctypes.data, n, default_version.encode(),
                                  default_scope.encode(), globals_pb or None, len(globals_pb), dict_slots, heap, C.byref(out))
    if rc < 0:
        raise RuntimeError(lib.hostsim_last_error().decode())
    if rc == 1:
        return 1, None
    w = WireBatch()
    w.n, w.n_tuples, w.n_roles, w.n_columns, w.dict_slots, w.K, w.fill_runs = out.n, out.n_tuples, out.n_roles, out.n_columns, out.dict_slots, out.K, out.fill_runs
    w.stats = dict(zip(STAT_NAMES, list(out.stats)))
    w.heap_len = min(out.heap_len, 1 << 30)
    w.req_u32 = _arr(out.req_u32, C.c_uint32, np.uint32, 16 * n).reshape(16, n)
    w.roles = _arr(out.roles, C.c_uint32, np.uint32, w.n_roles)
    w.tuple_action = _arr(out.tuple_action, C.c_uint32, np.uint32, w.n_tuples)
    w.col_tag = _arr(out.col_tag, C.c_uint8, np.uint8, w.n_columns * n).reshape(w.n_columns, n)
    w.col_val = _arr(out.col_val, C.c_uint64, np.uint64, w.n_columns * n).reshape(w.n_columns, n)
    w.heap_tag = _arr(out.heap_tag, C.c_uint8, np.uint8, w.heap_len)
    w.heap_val = _arr(out.heap_val, C.c_uint64, np.uint64, w.heap_len)
    w.dict = _arr(out.dict, C.c_uint64, np.uint64, w.dict_slots)
    w.dict_flags = _arr(out.dict_flags, C.c_uint32, np.uint32, w.dict_slots // 4 + 1).view(np.uint8)[:w.dict_slots].copy()
    w.in_span = _arr(out.in_span, C.c_uint32, np.uint32, n * 12).reshape(n, 12)
    w.act_span = _arr(out.act_span, C.c_uint32, np.uint32, w.n_tuples * 2).reshape(w.n_tuples, 2)
    total = int(offsets[-1]) if n else 0
    w.msg = _arr(out.msg, C.c_uint8, np.uint8, total + len(default_version.encode()) + len(default_scope.encode()) + 6 + len(globals_pb) + 8).tobytes()
    w.status = _arr(out.status, C.c_uint8, np.uint8, n)
    w.grouped = None   # CBH_WIRE_GROUP=1: (request words, column tags, column values) in grouped order, input -> position, routes
    if out.inv:
        w.grouped = (_arr(out.req_grouped, C.c_uint32, np.uint32, 16 * n).reshape(16, n),
                     _arr(out.col_tag_grouped, C.c_uint8, np.uint8, w.n_columns * n).reshape(w.n_columns, n),
                     _arr(out.col_val_grouped, C.c_uint64, np.uint64, w.n_columns * n).reshape(w.n_columns, n),
                     _arr(out.inv, C.c_uint32, np.uint32, n), int(out.n_routes))
    return 0, w


class Resolver:
    """string id -> bytes for either kind of batch"""

    def __init__(self, lt, host_batch=None, wire_batch=None):
        self.table = [s.encode() if isinstance(s, str) else bytes(s) for s in lt.strings]
        self.K = len(self.table)
        self.hb, self.wb = host_batch, wire_batch
        if host_batch is not None:
            self.raw = host_batch.str_bytes.tobytes()

    def __call__(self, sid):
        sid = int(sid)
        if sid < self.K:
            return self.table[sid]
        i = sid - self.K
        if self.hb is not None:
            return self.raw[int(self.hb.str_off[i]):int(self.hb.str_off[i + 1])]
        key = int(self.wb.dict[i])
        assert key != 0, "id of an empty dictionary slot"
        off, ln = key & 0xFFFFFFFF, (key >> 32) & 0xFFFF
        return self.wb.msg[off:off + ln]


def decode_value(tag, val, heap_tag, heap_val, res, depth=0):
    tag, val = int(tag), int(val)
    if tag == T_STRING:
        return ("s", res(val))
    if tag in (T_LIST, T_MAP):
        assert (val >> 62) == 1, "batch containers live in the batch heap"
        off, n = (val >> 32) & 0x3FFFFFFF, val & 0xFFFFFFFF
        cnt = 2 * n if tag == T_MAP else n
        items = tuple(decode_value(heap_tag[off + k], heap_val[off + k], heap_tag, heap_val, res, depth + 1) for k in range(cnt))
        return ("m" if tag == T_MAP else "l", items)
    return (tag, val)


def local_flags(lt, host_batch=None, wire_batch=None):
    """bytes -> CBH_SF_* of the batch-local strings that carry any"""
    out = {}
    if host_batch is not None:
        res = Resolver(lt, host_batch=host_batch)
        for i in range(host_batch.n_strings):
            if host_batch.str_flags[i]:
                out[res(res.K + i)] = int(host_batch.str_flags[i])
    else:
        res = Resolver(lt, wire_batch=wire_batch)
        for i in np.nonzero(wire_batch.dict_flags)[0]:
            out[res(res.K + int(i))] = int(wire_batch.dict_flags[i])
    return out


def assert_same_requests(lt, hb, wb, reads_request_strings):
    """host batch (cbi_flatten_pb, sort = 0) vs device batch"""
    n = wb.n
    assert hb.n_requests == n and hb.n_tuples == wb.n_tuples and len(hb.roles) == wb.n_roles
    hr, wr = Resolver(lt, host_batch=hb), Resolver(lt, wire_batch=wb)
    for f in RQ_PLAIN_FIELDS:
        assert np.array_equal(hb.req_u32[f], wb.req_u32[f]), "request field %d" % f
    for f in (6, 8):   # role / action offsets: both sides lay the slices out in input order
        assert np.array_equal(hb.req_u32[f], wb.req_u32[f]), "request field %d" % f
    fields = RQ_STRING_FIELDS + (RQ_RAW_STRING_FIELDS if reads_request_strings else ())
    for f in fields:
        for r in range(n):
            assert hr(hb.req_u32[f, r]) == wr(wb.req_u32[f, r]), "request %d field %d: %r != %r" % (r, f, hr(hb.req_u32[f, r]), wr(wb.req_u32[f, r]))
    assert [hr(x) for x in hb.roles] == [wr(x) for x in wb.roles]
    assert [hr(x) for x in hb.tuple_action] == [wr(x) for x in wb.tuple_action]
    assert np.array_equal(hb.col_tag, wb.col_tag), "column tags"
    for c in range(wb.n_columns):
        for r in range(n):
            a = decode_value(hb.col_tag[c, r], hb.col_val[c, r], hb.heap_tag, hb.heap_val, hr)
            b = decode_value(wb.col_tag[c, r], wb.col_val[c, r], wb.heap_tag, wb.heap_val, wr)
            assert a == b, "column %d request %d: %r != %r" % (c, r, a, b)
    assert local_flags(lt, host_batch=hb) == local_flags(lt, wire_batch=wb), "string usage flags"
    # identity of ids: equal strings <=> equal ids, on the device side too (the decision kernels compare ids)
    seen = {}
    for arr in (wb.req_u32[list(fields)].ravel(), wb.roles, wb.tuple_action):
        for x in arr:
            s = wr(x)
            assert seen.setdefault(s, int(x)) == int(x), "two ids for %r" % s


def to_batch(lt, wb, grouped=False):
    """``grouped``: the per-request arrays in the order the routing kernels left them (wb.grouped).
    A flatten.Batch the simulator's decision kernels accept: the dictionary made dense (test side only - the library hands the
    dictionary itself to the kernels, BatchDev.str_keys)."""
    b = Batch()
    b.n_requests, b.n_tuples = wb.n, wb.n_tuples
    used = np.nonzero(wb.dict)[0]
    remap = {int(wb.K + s): int(wb.K + k) for k, s in enumerate(used)}
    parts, off = [], [0]
    for s in used:
        key = int(wb.dict[s]); o, ln = key & 0xFFFFFFFF, (key >> 32) & 0xFFFF
        parts.append(wb.msg[o:o + ln]); off.append(off[-1] + ln)
    b.n_strings = len(used)
    b.str_off = np.array(off, dtype=np.uint32)
    b.str_bytes = np.frombuffer(b"".join(parts), dtype=np.uint8).copy() if parts else np.zeros(0, np.uint8)
    b.str_flags = wb.dict_flags[used].astype(np.uint8)
    rm = np.vectorize(lambda x: remap.get(int(x), int(x)), otypes=[np.uint32])
    b.req_u32 = (wb.grouped[0] if grouped else wb.req_u32).copy()
    for f in RQ_STRING_FIELDS + RQ_RAW_STRING_FIELDS:
        b.req_u32[f] = rm(b.req_u32[f]) if wb.n else b.req_u32[f]
    b.roles = rm(wb.roles) if wb.n_roles else wb.roles.copy()
    b.tuple_action = rm(wb.tuple_action) if wb.n_tuples else wb.tuple_action.copy()
    b.tuple_req = np.repeat(np.arange(wb.n, dtype=np.uint32), wb.req_u32[9].astype(np.int64)) if not grouped else np.zeros(wb.n_tuples, np.uint32)
    b.col_tag = (wb.grouped[1] if grouped else wb.col_tag).copy()
    b.col_val = (wb.grouped[2] if grouped else wb.col_val).copy()
    m = b.col_tag == T_STRING
    if m.any():
        b.col_val[m] = rm(b.col_val[m]).astype(np.uint64)
    b.heap_tag = wb.heap_tag.copy()
    b.heap_val = wb.heap_val.copy()
    hm = b.heap_tag == T_STRING
    if hm.any():
        b.heap_val[hm] = rm(b.heap_val[hm]).astype(np.uint64)
    b.tuple_perm = np.arange(wb.n_tuples, dtype=np.int64)
    b.req_perm = None
    return b


def sim_outputs(lt, res, n, cap=None, edr_is_grouped=False):
    """The device assembler (cbh_wire_out_* kernels, simulator) on the batch the LAST sim_flatten call built; ``res`` = a
    capi.Result in input order.  -> ([serialized CheckOutput], flags uint8[n])"""
    lib = hostsim_api.lib()
    lib.hostsim_wire_outputs.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    lib.hostsim_wire_outputs.restype = C.c_longlong
    buf = C.create_string_buffer(lt.blob, len(lt.blob))
    cap = 256 if cap is None else cap
    while True:
        out = np.zeros(cap + 8, np.uint8)
        off = np.zeros(n + 1, np.uint64)
        flags = np.zeros(n + 1, np.uint8)
        need = lib.hostsim_wire_outputs(C.cast(buf, C.c_void_p), len(lt.blob), res.effect.ctypes.data, res.policy.ctypes.data, res.scope.ctypes.data,
                                        res.status.ctypes.data, res.edr.ctypes.data, out.ctypes.data, cap, off.ctypes.data, flags.ctypes.data, int(edr_is_grouped))
        if need < 0:
            raise RuntimeError(lib.hostsim_last_error().decode())
        if need <= cap:
            raw = out.tobytes()
            assert int(off[n]) == need
            return [raw[int(off[i]):int(off[i + 1])] for i in range(n)], flags[:n]
        cap = int(need)
