"""ctypes binding of ``libcerbos_ingest.so`` (include/cerbos_ingest.h): serialized ``CheckInput`` messages ->
``flatten.Batch``, by the C++ ingest instead of the Python flattener.

The C++ ingest and ``flatten.Flattener`` implement the same contract (tests/test_ingest.py compares them
array for array); this module exists so Python callers, the tests and the bench can drive the native one.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi
from .flatten import RQ_NFIELDS, Batch

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcerbos_ingest.so")
_lib = None


class IngestError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IngestError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        lib.cbi_last_error.restype = C.c_char_p
        lib.cbi_table_open.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        lib.cbi_table_close.argtypes = [vp]
        lib.cbi_table_close.restype = None
        lib.cbi_flatten_pb.argtypes = [vp, vp, vp, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
        lib.cbi_flatten_pb_mt.argtypes = [vp, vp, vp, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
        lib.cbi_flatten_pb_g.argtypes = [vp, vp, vp, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(vp)]
        lib.cbi_flatten_request_pb.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
        lib.cbi_assemble_response_pb.argtypes = [vp, vp, C.POINTER(capi.CResult), vp, C.c_uint64, C.c_char_p, C.POINTER(vp)]
        lib.cbi_batch_free.argtypes = [vp]
        lib.cbi_batch_free.restype = None
        lib.cbi_batch_view.argtypes = [vp]
        lib.cbi_batch_view.restype = C.POINTER(capi.CBatch)
        lib.cbi_batch_tuple_perm.argtypes = [vp]
        lib.cbi_batch_tuple_perm.restype = C.POINTER(C.c_uint64)
        lib.cbi_batch_request_input.argtypes = [vp]
        lib.cbi_batch_request_input.restype = C.POINTER(C.c_uint32)
        lib.cbi_assemble_pb.argtypes = [vp, vp, C.POINTER(capi.CResult), vp, vp, C.c_uint32, C.c_char_p, C.POINTER(vp)]
        lib.cbi_assemble_pb_mt.argtypes = [vp, vp, C.POINTER(capi.CResult), vp, vp, C.c_uint32, C.c_char_p, C.c_int, C.POINTER(vp)]
        lib.cbi_assemble_wire_pb.argtypes = [vp, C.POINTER(capi.CResult), vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_char_p, C.c_int, C.POINTER(vp)]
        lib.cbi_table_trace_scope.argtypes = [vp]
        lib.cbi_table_trace_scope.restype = C.c_uint32
        lib.cbi_trace_pb.argtypes = [vp, vp, C.POINTER(capi.CResult), vp, C.c_uint32, vp, vp, C.c_uint32, C.POINTER(vp)]
        lib.cbi_trace_request_pb.argtypes = [vp, vp, C.POINTER(capi.CResult), vp, C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(vp)]
        lib.cbi_assemble_response_traced_pb.argtypes = [vp, vp, C.POINTER(capi.CResult), vp, C.c_uint64, C.c_char_p, vp, C.POINTER(vp)]
        lib.cbi_outputs_free.argtypes = [vp]
        lib.cbi_outputs_free.restype = None
        lib.cbi_outputs_bytes.argtypes = [vp]
        lib.cbi_outputs_bytes.restype = C.POINTER(C.c_uint8)
        lib.cbi_outputs_offsets.argtypes = [vp]
        lib.cbi_outputs_offsets.restype = C.POINTER(C.c_uint64)
        lib.cbi_outputs_flags.argtypes = [vp]
        lib.cbi_outputs_flags.restype = C.POINTER(C.c_uint8)
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise IngestError(load().cbi_last_error().decode("utf-8", "replace"))


def _copy(ptr, ctype, dtype, n):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True)


class _NativeBatch:
    """Keeps a ``cbi_batch`` alive for response assembly."""

    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                load().cbi_batch_free(self.h)
                self.h = None
        except Exception:
            pass


class IngestTable:
    """Host dictionaries of one lowered table image (``cbi_table``)."""

    def __init__(self, blob: bytes):
        self.h = C.c_void_p()
        self._blob = bytes(blob)
        _check(load().cbi_table_open(self._blob, len(self._blob), C.byref(self.h)))

    def close(self):
        if self.h:
            load().cbi_table_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace_scope(self) -> int:
        """0 = no trace sections, 1 = trace the inputs marked CEL_ERROR, 2 = trace every input (``cbi_table_trace_scope``)."""
        return int(load().cbi_table_trace_scope(self.h))

    def flatten_pb(self, data, offsets, default_policy_version="default", default_scope="", sort=True, threads=1, globals_pb=b"") -> Batch:
        """``data``: uint8 array holding the messages back to back, ``offsets``: uint64[n + 1]; ``threads`` > 1
        flattens slices concurrently inside the call (same batch, bit for bit)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        h = C.c_void_p()
        _check(load().cbi_flatten_pb_g(self.h, data.ctypes.data if data.size else None, offsets.ctypes.data, n,
                                       default_policy_version.encode(), default_scope.encode(), globals_pb or None, len(globals_pb or b""),
                                       int(bool(sort)), int(threads), C.byref(h)))
        return self._batch(h)

    def flatten_request_pb(self, request: bytes, aux_data: bytes = None, default_policy_version="default", default_scope="",
                           sort=True, threads=1) -> Batch:
        """One serialized ``CheckResourcesRequest`` (+ the serialized engine ``AuxData`` the server derived from its
        JWT, if any): one device request per resource entry."""
        h = C.c_void_p()
        _check(load().cbi_flatten_request_pb(self.h, request, len(request), aux_data, len(aux_data) if aux_data else 0,
                                             default_policy_version.encode(), default_scope.encode(), int(bool(sort)),
                                             int(threads), C.byref(h)))
        return self._batch(h)

    def assemble_response_pb(self, batch, res, request: bytes, default_policy_version="default", traced=None, aux_data: bytes = None):
        """-> (serialized ``CheckResourcesResponse``, flags uint8[n resource entries]).  ``traced`` = (device-order Result,
        records) of ``capi.Table.trace`` on the SAME batch: its outputs go into ``ResultEntry.outputs`` (cbi_trace_request_pb,
        cbi_assemble_response_traced_pb)."""
        h = C.c_void_p()
        th = C.c_void_p()
        if traced is not None:
            tres, records = traced
            records = np.ascontiguousarray(records, dtype=np.uint32)
            _check(load().cbi_trace_request_pb(self.h, batch.native.h, C.byref(tres.c), records.ctypes.data if records.size else None,
                                               len(records), request, len(request), aux_data, len(aux_data) if aux_data else 0, C.byref(th)))
        try:
            _check(load().cbi_assemble_response_traced_pb(self.h, batch.native.h, C.byref(res.c), request, len(request),
                                                          default_policy_version.encode(), th if traced is not None else None, C.byref(h)))
        finally:
            if th:
                load().cbi_outputs_free(th)
        try:
            off = _copy(load().cbi_outputs_offsets(h), C.c_uint64, np.int64, 2)
            raw = _copy(load().cbi_outputs_bytes(h), C.c_uint8, np.uint8, int(off[-1])).tobytes()
            n_entries = int(np.unique(batch.vreq_input).size) if batch.vreq_input.size else 0
            flags = _copy(load().cbi_outputs_flags(h), C.c_uint8, np.uint8, n_entries)
            return raw, flags
        finally:
            load().cbi_outputs_free(h)

    def _batch(self, h) -> Batch:
        try:
            v = load().cbi_batch_view(h).contents
            b = Batch()
            R, T = v.n_requests, v.n_tuples
            b.n_requests, b.n_tuples, b.n_strings = R, T, v.n_strings
            b.req_u32 = _copy(v.req_u32, C.c_uint32, np.uint32, RQ_NFIELDS * R).reshape(RQ_NFIELDS, R)
            b.roles = _copy(v.roles, C.c_uint32, np.uint32, v.n_roles)
            b.tuple_req = _copy(v.tuple_req, C.c_uint32, np.uint32, T)
            b.tuple_action = _copy(v.tuple_action, C.c_uint32, np.uint32, T)
            b.col_tag = _copy(v.col_tag, C.c_uint8, np.uint8, v.n_columns * R).reshape(v.n_columns, R)
            b.col_val = _copy(v.col_val, C.c_uint64, np.uint64, v.n_columns * R).reshape(v.n_columns, R)
            b.heap_tag = _copy(v.heap_tag, C.c_uint8, np.uint8, v.heap_len)
            b.heap_val = _copy(v.heap_val, C.c_uint64, np.uint64, v.heap_len)
            b.str_off = _copy(v.str_off, C.c_uint32, np.uint32, v.n_strings + 1)
            b.str_bytes = _copy(v.str_bytes, C.c_uint8, np.uint8, v.str_bytes_len)
            b.str_flags = _copy(v.str_flags, C.c_uint8, np.uint8, v.n_strings)
            b.tuple_perm = _copy(load().cbi_batch_tuple_perm(h), C.c_uint64, np.int64, T)
            b.req_perm = None
            b.vreq_input = _copy(load().cbi_batch_request_input(h), C.c_uint32, np.int64, R)   # device request -> input
            b.native = _NativeBatch(h)
            return b
        except Exception:
            load().cbi_batch_free(h)
            raise

    def assemble_pb(self, batch, res, data, offsets, default_policy_version="default", threads=1):
        """Device-order results (``capi.Result`` before ``to_input_order``) of a batch made by ``flatten_pb`` from
        the same messages -> ([serialized CheckOutput], flags uint8[n])."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        h = C.c_void_p()
        _check(load().cbi_assemble_pb_mt(self.h, batch.native.h, C.byref(res.c), data.ctypes.data if data.size else None,
                                         offsets.ctypes.data, n, default_policy_version.encode(), int(threads), C.byref(h)))
        try:
            off = _copy(load().cbi_outputs_offsets(h), C.c_uint64, np.int64, n + 1)
            raw = _copy(load().cbi_outputs_bytes(h), C.c_uint8, np.uint8, int(off[-1])).tobytes()
            flags = _copy(load().cbi_outputs_flags(h), C.c_uint8, np.uint8, n)
            return [raw[off[i]:off[i + 1]] for i in range(n)], flags
        finally:
            load().cbi_outputs_free(h)


    def assemble_wire_pb(self, res, data, offsets, spans, default_policy_version="default", threads=1):
        """Results of a batch the device flattened (``capi.Table.wire_flatten``; ``spans`` = its ``spans()``) ->
        ([serialized CheckOutput], flags uint8[n])."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        in_span, act_span, act_off = spans
        n = len(offsets) - 1
        h = C.c_void_p()
        _check(load().cbi_assemble_wire_pb(self.h, C.byref(res.c), data.ctypes.data if data.size else None, offsets.ctypes.data, n,
                                           int(act_off[-1]) if n else 0, in_span.ctypes.data, act_span.ctypes.data, act_off.ctypes.data,
                                           default_policy_version.encode(), int(threads), C.byref(h)))
        return self._outputs(h, n)

    def _outputs(self, h, n):
        try:
            off = _copy(load().cbi_outputs_offsets(h), C.c_uint64, np.int64, n + 1)
            raw = _copy(load().cbi_outputs_bytes(h), C.c_uint8, np.uint8, int(off[-1])).tobytes()
            flags = _copy(load().cbi_outputs_flags(h), C.c_uint8, np.uint8, n)
            return [raw[off[i]:off[i + 1]] for i in range(n)], flags
        finally:
            load().cbi_outputs_free(h)


TRACE_ERRORS_INCOMPLETE, TRACE_OUTPUTS_INCOMPLETE = 4, 8


def trace_pb(table: IngestTable, batch, res, records, data, offsets):
    """``cbi_trace_pb``: the log of a traced batch (``capi.Table.trace``: device-order ``res``, ``records``) ->
    ([the evaluation_errors + outputs fields of each input's CheckOutput, serialized], flags uint8[n])."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    records = np.ascontiguousarray(records, dtype=np.uint32)
    n = len(offsets) - 1
    h = C.c_void_p()
    _check(load().cbi_trace_pb(table.h, batch.native.h, C.byref(res.c), records.ctypes.data if records.size else None,
                               len(records), data.ctypes.data if data.size else None, offsets.ctypes.data, n, C.byref(h)))
    return table._outputs(h, n)


class WireFlattener:
    """Drop-in for ``flatten.Flattener`` that goes through the wire format and the C++ ingest:
    dict -> ``CheckInput`` bytes (wire.py) -> ``cbi_flatten_pb``."""

    def __init__(self, lt):
        self.table = IngestTable(lt.blob)

    def flatten(self, inputs, default_policy_version="default", default_scope="", sort=True, globals_=None) -> Batch:
        from . import wire
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
        b = self.table.flatten_pb(data, off, default_policy_version, default_scope, sort,
                                  globals_pb=wire.encode_map(1, globals_) if globals_ else b"")
        b.actions_per_request = [list(inp.get("actions") or []) for inp in inputs]
        return b
