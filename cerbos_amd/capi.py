"""ctypes binding of libcerbos_hip.so (include/cerbos_hip.h).

This is the stand-in for the cgo binding shown in INTEGRATION.md: same entry points, same
structs.  There is no CPU fallback: if the library is missing, or no MI355X is visible,
``load()`` / ``init()`` raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcerbos_hip.so")

ABI_VERSION = 3
MAX_DEVICES = 16
F_LENIENT_SCOPE_SEARCH = 1
F_STRICT_EVALUATION = 2
F_WANT_DERIVED_ROLES = 4
F_WANT_EFFECTIVE_POLICIES = 8

EFFECT_ALLOW, EFFECT_DENY = 1, 2
ST_OK, ST_CEL_ERROR, ST_UNSUPPORTED, ST_WANTS_TRACE = 0, 1, 2, 3
P_EMPTY, P_NO_MATCH, P_RESOURCE, P_PRINCIPAL, P_TABLE, P_NO_MATCH_SP = range(6)
NONE = 0xFFFFFFFF

EXPORTED_SYMBOLS = [
    "cbh_init", "cbh_shutdown", "cbh_last_error", "cbh_abi_version", "cbh_num_devices", "cbh_device_ordinal",
    "cbh_alloc_pinned", "cbh_free_pinned", "cbh_batch_slab_bytes", "cbh_batch_bind_slab", "cbh_result_slab_bytes", "cbh_result_bind_slab",
    "cbh_table_load", "cbh_table_retain", "cbh_table_release", "cbh_table_broadcast_kind", "cbh_table_num_strings", "cbh_table_num_columns",
    "cbh_table_device_bytes", "cbh_table_device_ptr", "cbh_table_adopt_device_image",
    "cbh_check_batch", "cbh_trace_batch", "cbh_batch_upload", "cbh_batch_upload_on", "cbh_batch_release", "cbh_check_resident",
    "cbh_synchronize", "cbh_result_download", "cbh_kernel_time_ms", "cbh_plan_describe",
    "cbh_check_resident_many", "cbh_table_set_resident_streams", "cbh_table_resident_streams",
    "cbh_wire_flatten", "cbh_wire_spans_download", "cbh_wire_outputs", "cbh_wire_check_pb", "cbh_wire_check_pb_submit", "cbh_wire_check_pb_collect",
    "cbh_wire_flatten_requests", "cbh_wire_check_requests_pb", "cbh_wire_check_requests_trail_pb", "cbh_batch_set_trail", "cbh_trail_download",
    "cbh_table_num_policies", "cbh_table_policy_key", "cbh_check_batch_trail",
]


class HipEngineError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_devices", C.c_uint32), ("devices", C.c_int32 * MAX_DEVICES)]


class CBatch(C.Structure):
    _fields_ = [
        ("n_requests", C.c_uint32), ("n_tuples", C.c_uint32), ("n_roles", C.c_uint32),
        ("n_columns", C.c_uint32), ("n_strings", C.c_uint32), ("heap_len", C.c_uint32),
        ("str_bytes_len", C.c_uint64),
        ("req_u32", C.c_void_p), ("roles", C.c_void_p), ("tuple_req", C.c_void_p),
        ("tuple_action", C.c_void_p), ("col_tag", C.c_void_p), ("col_val", C.c_void_p),
        ("heap_tag", C.c_void_p), ("heap_val", C.c_void_p), ("str_off", C.c_void_p),
        ("str_bytes", C.c_void_p), ("str_flags", C.c_void_p),
    ]


class CParams(C.Structure):
    _fields_ = [("now_ns", C.c_int64), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class CResult(C.Structure):
    _fields_ = [("effect", C.c_void_p), ("policy", C.c_void_p), ("scope", C.c_void_p),
                ("status", C.c_void_p), ("edr_mask", C.c_void_p)]


class CTrace(C.Structure):
    _fields_ = [("records", C.c_void_p), ("capacity", C.c_uint32), ("count", C.c_uint32)]


class CWireInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_requests", "n_tuples", "n_host", "first_bad", "dict_slots", "heap_len", "fill_runs", "n_routes")]


class HostFlattenerNeeded(RuntimeError):
    """cbh_wire_flatten returned 1: these messages are the host flattener's (libcerbos_ingest.so) - same results, other road."""


TRACE_RECORD_WORDS = 8

_lib = None


def load():
    """dlopen the in-tree library (built by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipEngineError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the decision path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    lib.cbh_init.argtypes = [C.POINTER(Config)]
    lib.cbh_init.restype = i32
    lib.cbh_shutdown.restype = None
    lib.cbh_last_error.restype = C.c_char_p
    lib.cbh_abi_version.restype = u32
    lib.cbh_table_load.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.cbh_table_load.restype = i32
    lib.cbh_table_release.argtypes = [vp]
    lib.cbh_table_release.restype = None
    lib.cbh_table_retain.argtypes = [vp]
    lib.cbh_table_retain.restype = None
    lib.cbh_table_broadcast_kind.argtypes = [vp]
    lib.cbh_table_broadcast_kind.restype = C.c_char_p
    lib.cbh_num_devices.restype = u32
    lib.cbh_device_ordinal.argtypes = [u32]
    lib.cbh_device_ordinal.restype = i32
    lib.cbh_alloc_pinned.argtypes = [C.c_size_t]
    lib.cbh_alloc_pinned.restype = vp
    lib.cbh_free_pinned.argtypes = [vp]
    lib.cbh_free_pinned.restype = None
    lib.cbh_batch_slab_bytes.argtypes = [C.POINTER(CBatch)]
    lib.cbh_batch_slab_bytes.restype = C.c_size_t
    lib.cbh_batch_bind_slab.argtypes = [C.POINTER(CBatch), vp]
    lib.cbh_batch_bind_slab.restype = None
    lib.cbh_result_slab_bytes.argtypes = [u32, u32]
    lib.cbh_result_slab_bytes.restype = C.c_size_t
    lib.cbh_result_bind_slab.argtypes = [C.POINTER(CResult), vp, u32, u32]
    lib.cbh_result_bind_slab.restype = None
    lib.cbh_batch_upload_on.argtypes = [vp, u32, C.POINTER(CBatch), C.POINTER(vp)]
    lib.cbh_batch_upload_on.restype = i32
    lib.cbh_table_num_strings.argtypes = [vp]
    lib.cbh_table_num_strings.restype = u32
    lib.cbh_table_num_columns.argtypes = [vp]
    lib.cbh_table_num_columns.restype = u32
    lib.cbh_table_device_bytes.argtypes = [vp]
    lib.cbh_table_device_bytes.restype = C.c_uint64
    lib.cbh_table_device_ptr.argtypes = [vp]
    lib.cbh_table_device_ptr.restype = vp
    lib.cbh_table_adopt_device_image.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.cbh_table_adopt_device_image.restype = i32
    lib.cbh_check_batch.argtypes = [vp, C.POINTER(CBatch), C.POINTER(CParams), C.POINTER(CResult)]
    lib.cbh_check_batch.restype = i32
    lib.cbh_trace_batch.argtypes = [vp, C.POINTER(CBatch), C.POINTER(CParams), C.POINTER(CResult), C.POINTER(CTrace)]
    lib.cbh_trace_batch.restype = i32
    lib.cbh_batch_upload.argtypes = [vp, C.POINTER(CBatch), C.POINTER(vp)]
    lib.cbh_batch_upload.restype = i32
    lib.cbh_batch_release.argtypes = [vp]
    lib.cbh_batch_release.restype = None
    lib.cbh_check_resident.argtypes = [vp, vp, C.POINTER(CParams)]
    lib.cbh_check_resident.restype = i32
    lib.cbh_synchronize.argtypes = [vp]
    lib.cbh_synchronize.restype = i32
    lib.cbh_result_download.argtypes = [vp, vp, C.POINTER(CResult)]
    lib.cbh_result_download.restype = i32
    lib.cbh_kernel_time_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.cbh_kernel_time_ms.restype = i32
    lib.cbh_check_resident_many.argtypes = [vp, C.POINTER(vp), u32, C.POINTER(CParams)]
    lib.cbh_check_resident_many.restype = i32
    lib.cbh_table_set_resident_streams.argtypes = [vp, u32]
    lib.cbh_table_set_resident_streams.restype = i32
    lib.cbh_table_resident_streams.argtypes = [vp]
    lib.cbh_table_resident_streams.restype = u32
    lib.cbh_wire_flatten.argtypes = [vp, u32, vp, vp, u32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(vp), C.POINTER(CWireInfo)]
    lib.cbh_wire_flatten.restype = i32
    lib.cbh_wire_spans_download.argtypes = [vp, vp, vp, vp, vp]
    lib.cbh_wire_spans_download.restype = i32
    lib.cbh_wire_outputs.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.POINTER(C.c_size_t)]
    lib.cbh_wire_outputs.restype = i32
    lib.cbh_wire_check_pb.argtypes = [vp, u32, vp, vp, u32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(CParams), vp, C.c_size_t, vp, vp,
                                      C.POINTER(C.c_size_t), C.POINTER(CWireInfo)]
    lib.cbh_wire_check_pb.restype = i32
    if hasattr(lib, "cbh_wire_check_pb_submit"):   # (a library of an earlier round in a same-box A/B: tests/test_abi.py is what insists on every symbol)
        lib.cbh_wire_check_pb_submit.argtypes = [vp, u32, vp, vp, u32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(CParams), vp, C.c_size_t, vp, vp, C.POINTER(vp)]
        lib.cbh_wire_check_pb_submit.restype = i32
        lib.cbh_wire_check_pb_collect.argtypes = [vp, vp, C.POINTER(C.c_size_t), C.POINTER(CWireInfo)]
        lib.cbh_wire_check_pb_collect.restype = i32
    lib.cbh_wire_flatten_requests.argtypes = [vp, u32, vp, vp, u32, vp, vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, vp, vp, C.POINTER(vp), C.POINTER(CWireInfo)]
    lib.cbh_wire_flatten_requests.restype = i32
    lib.cbh_wire_check_requests_pb.argtypes = [vp, u32, vp, vp, u32, vp, vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(CParams), vp, vp,
                                               vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(CWireInfo)]
    lib.cbh_wire_check_requests_pb.restype = i32
    lib.cbh_wire_check_requests_trail_pb.argtypes = lib.cbh_wire_check_requests_pb.argtypes + [vp]
    lib.cbh_wire_check_requests_trail_pb.restype = i32
    lib.cbh_table_num_policies.argtypes = [vp]
    lib.cbh_table_num_policies.restype = u32
    lib.cbh_table_policy_key.argtypes = [vp, u32, C.POINTER(C.c_char_p), C.POINTER(u32)]
    lib.cbh_table_policy_key.restype = i32
    lib.cbh_batch_set_trail.argtypes = [vp, vp, vp, u32]
    lib.cbh_batch_set_trail.restype = i32
    lib.cbh_trail_download.argtypes = [vp, vp, vp]
    lib.cbh_trail_download.restype = i32
    lib.cbh_check_batch_trail.argtypes = [vp, C.POINTER(CBatch), C.POINTER(CParams), C.POINTER(CResult), vp, u32, vp]
    lib.cbh_check_batch_trail.restype = i32
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise HipEngineError(load().cbh_last_error().decode("utf-8", "replace"))


_inited_device = None


def init(device=0):
    """``device``: one HIP ordinal, or the list of ordinals the engine may use (cbh_config.devices; a batch is then
    sharded over them and the table image broadcast once).  ``"all"`` = every visible device."""
    global _inited_device
    lib = load()
    if lib.cbh_abi_version() != ABI_VERSION:
        raise HipEngineError("libcerbos_hip.so ABI mismatch")
    devs = [] if device == "all" else ([int(device)] if isinstance(device, int) else [int(d) for d in device])
    cfg = Config(ABI_VERSION, len(devs), (C.c_int32 * MAX_DEVICES)(*devs))
    _check(lib.cbh_init(C.byref(cfg)))
    _inited_device = devs[0] if devs else 0


def num_devices() -> int:
    return load().cbh_num_devices()


class _PinnedBlock:
    """Owner of one cbh_alloc_pinned block; numpy views keep it alive through their ``base`` chain."""

    def __init__(self, nbytes):
        self.ptr = load().cbh_alloc_pinned(max(1, nbytes))
        if not self.ptr:
            raise HipEngineError("cbh_alloc_pinned failed")
        self.nbytes = nbytes

    def __del__(self):
        try:
            if self.ptr:
                load().cbh_free_pinned(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (cbh_alloc_pinned): cbh_check_batch moves such arrays by DMA."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) if not isinstance(shape, int) else int(shape)
    blk = _PinnedBlock(n * dtype.itemsize)
    buf = (C.c_uint8 * max(1, n * dtype.itemsize)).from_address(blk.ptr)
    buf._owner = blk   # the ctypes buffer is the numpy array's base: keeps the block alive
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def make_cbatch(batch, n_columns):
    """flatten.Batch -> CBatch (keeps the numpy arrays alive through the returned tuple)."""
    cb = CBatch()
    cb.n_requests = batch.n_requests
    cb.n_tuples = batch.n_tuples
    cb.n_roles = int(batch.roles.size)
    cb.n_columns = n_columns
    cb.n_strings = batch.n_strings
    cb.heap_len = int(batch.heap_tag.size)
    cb.str_bytes_len = int(batch.str_bytes.size)
    for f in ("req_u32", "roles", "tuple_req", "tuple_action", "col_tag", "col_val", "heap_tag",
              "heap_val", "str_off", "str_bytes", "str_flags"):
        setattr(cb, f, _ptr(getattr(batch, f)))
    return cb


def _slab_view(slab, addr, shape, dtype):
    """numpy view of `shape` x `dtype` at absolute address `addr` inside the pinned uint8 array `slab`."""
    off = addr - slab.ctypes.data
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return slab[off:off + n].view(dtype).reshape(shape)


def pin_batch(batch):
    """Move a flatten.Batch's arrays into ONE page-locked slab in the library's canonical order
    (cbh_batch_bind_slab): cbh_check_batch then uploads the batch with a single DMA.  In place; returns the batch."""
    lib = load()
    n_columns = int(batch.col_tag.shape[0]) if batch.col_tag.ndim == 2 else 0
    cb = make_cbatch(batch, n_columns)
    slab = pinned_empty(max(1, lib.cbh_batch_slab_bytes(C.byref(cb))), np.uint8)
    lib.cbh_batch_bind_slab(C.byref(cb), slab.ctypes.data_as(C.c_void_p))
    for f in ("req_u32", "roles", "tuple_action", "col_tag", "col_val", "heap_tag", "heap_val", "str_off", "str_bytes", "str_flags"):
        a = getattr(batch, f)
        if a is not None and a.size:
            v = _slab_view(slab, getattr(cb, f), a.shape, a.dtype)
            v[...] = a
            setattr(batch, f, v)
    batch.slab = slab
    return batch


class Table:
    def __init__(self, blob: bytes):
        if _inited_device is None:
            init(0)
        lib = load()
        h = C.c_void_p()
        buf = C.create_string_buffer(blob, len(blob))
        _check(lib.cbh_table_load(C.cast(buf, C.c_void_p), len(blob), C.byref(h)))
        self.h = h
        self.num_strings = lib.cbh_table_num_strings(h)
        self.num_columns = lib.cbh_table_num_columns(h)

    @classmethod
    def adopt(cls, device_ptr: int, length: int):
        lib = load()
        self = cls.__new__(cls)
        h = C.c_void_p()
        _check(lib.cbh_table_adopt_device_image(C.c_void_p(device_ptr), length, C.byref(h)))
        self.h = h
        self.num_strings = lib.cbh_table_num_strings(h)
        self.num_columns = lib.cbh_table_num_columns(h)
        return self

    @classmethod
    def borrow(cls, other: "Table"):
        """A second handle object on ``other``'s table holding its own reference (cbh_table_retain): stays valid
        after ``other.close()``; its own ``close()`` drops that reference."""
        load().cbh_table_retain(other.h)
        self = cls.__new__(cls)
        self.h = C.c_void_p(other.h.value)
        self.num_strings, self.num_columns = other.num_strings, other.num_columns
        return self

    def device_ptr(self) -> int:
        return load().cbh_table_device_ptr(self.h)

    def device_bytes(self) -> int:
        return load().cbh_table_device_bytes(self.h)

    def close(self):
        if getattr(self, "h", None):
            load().cbh_table_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one-shot
    def check(self, batch, now_ns=0, flags=0, want=("policy", "scope", "status", "edr"), device_order=False, pinned=False,
              into=None):
        """``pinned``: results in page-locked memory (with a pin_batch'ed batch the call is pure DMA);
        ``into``: reuse a Result of the right shape instead of allocating one."""
        res = into if into is not None else Result(batch.n_tuples, batch.n_requests, want, pinned)
        cb = make_cbatch(batch, self.num_columns)
        p = CParams(now_ns, flags, 0)
        _check(load().cbh_check_batch(self.h, C.byref(cb), C.byref(p), C.byref(res.c)))
        return res if device_order else res.to_input_order(batch)

    def check_trail(self, batch, groups=None, n_groups=1, now_ns=0, flags=0, want=("policy", "scope", "status", "edr")):
        """``cbh_check_batch_trail``: the decisions (DEVICE order) and, per group of requests, the mask of the policies whose bindings
        the walk iterated (AuditTrail.EffectivePolicies) -> (Result, uint32[n_groups][words]).  ``groups``: the group of every request
        of ``batch`` in its (device) order, None = one group."""
        res = Result(batch.n_tuples, batch.n_requests, want)
        cb = make_cbatch(batch, self.num_columns)
        p = CParams(now_ns, flags, 0)
        words = (int(load().cbh_table_num_policies(self.h)) + 31) // 32
        masks = np.zeros((max(n_groups, 1), max(words, 1)), dtype=np.uint32)
        grp = None if groups is None else np.ascontiguousarray(groups, dtype=np.uint32)
        _check(load().cbh_check_batch_trail(self.h, C.byref(cb), C.byref(p), C.byref(res.c), grp.ctypes.data if grp is not None else None,
                                            n_groups, masks.ctypes.data))
        return res, masks[:, :words]

    def policy_keys(self):
        """``cbh_table_policy_key`` for every policy of the table (the bit positions of ``check_trail``'s masks)."""
        out = []
        for i in range(int(load().cbh_table_num_policies(self.h))):
            key, ln = C.c_char_p(), C.c_uint32()
            _check(load().cbh_table_policy_key(self.h, i, C.byref(key), C.byref(ln)))
            out.append(C.string_at(key, ln.value).decode("utf-8"))
        return out

    def trace(self, batch, now_ns=0, flags=0, capacity=None):
        """``cbh_trace_batch``: decide ``batch`` with the tracing kernel -> (Result in DEVICE order, records uint32[n][8]).
        The log is sized from the batch and grown until it holds every record."""
        cap = capacity or max(256, 4 * batch.n_tuples)
        cb = make_cbatch(batch, self.num_columns)
        p = CParams(now_ns, flags, 0)
        while True:
            res = Result(batch.n_tuples, batch.n_requests, ("policy", "scope", "status", "edr"))
            rec = np.zeros((cap, TRACE_RECORD_WORDS), dtype=np.uint32)
            tr = CTrace(rec.ctypes.data, cap, 0)
            _check(load().cbh_trace_batch(self.h, C.byref(cb), C.byref(p), C.byref(res.c), C.byref(tr)))
            if tr.count <= cap:
                return res, rec[:tr.count]
            cap = int(tr.count) + 64

    # ---- resident
    def broadcast_kind(self) -> str:
        return load().cbh_table_broadcast_kind(self.h).decode()

    def retain(self):
        load().cbh_table_retain(self.h)

    def upload(self, batch, device_index=0):
        cb = make_cbatch(batch, self.num_columns)
        h = C.c_void_p()
        _check(load().cbh_batch_upload_on(self.h, device_index, C.byref(cb), C.byref(h)))
        return DeviceBatch(self, h, batch.n_tuples, batch.n_requests, batch)

    def wire_flatten(self, data, offsets, default_policy_version="default", default_scope="", device_index=0, globals_pb=b""):
        """``cbh_wire_flatten``: serialized CheckInputs (uint8 array + uint64[n + 1] offsets) -> a resident batch the GPU
        flattened.  Results of ``launch`` + ``download`` on it are in input order.  Raises ``HostFlattenerNeeded`` when the
        messages are the host flattener's.  ``globals_pb``: the call's globals as a serialized google.protobuf.Struct (a table
        lowered with per-call globals reads them)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        h = C.c_void_p()
        info = CWireInfo()
        rc = load().cbh_wire_flatten(self.h, device_index, data.ctypes.data if data.size else None, offsets.ctypes.data, n,
                                     default_policy_version.encode(), default_scope.encode(), globals_pb or None, len(globals_pb or b""),
                                     C.byref(h), C.byref(info))
        if rc == 1:
            raise HostFlattenerNeeded(load().cbh_last_error().decode("utf-8", "replace"))
        _check(rc)
        db = DeviceBatch(self, h, info.n_tuples, n)
        db.wire_info = {f[0]: getattr(info, f[0]) for f in CWireInfo._fields_}
        return db

    def wire_check_pb(self, data, offsets, now_ns=0, flags=0, default_policy_version="default", default_scope="", device_index=0,
                      globals_pb=b"", out=None):
        """``cbh_wire_check_pb``: the device road in one call.  -> ([serialized CheckOutput], flags uint8[n]).  ``out`` =
        (bytes uint8[cap], offsets uint64[n + 1], flags uint8[n]) to reuse (page-locked) buffers; they are grown when short."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        ob, oo, of = out if out is not None else (np.empty(256 * n + 4096, dtype=np.uint8), np.empty(n + 1, dtype=np.uint64), np.empty(max(n, 1), dtype=np.uint8))
        p = CParams(now_ns, flags, 0)
        info, need = CWireInfo(), C.c_size_t()
        for _ in range(2):
            rc = load().cbh_wire_check_pb(self.h, device_index, data.ctypes.data if data.size else None, offsets.ctypes.data, n,
                                          default_policy_version.encode(), default_scope.encode(), globals_pb or None, len(globals_pb or b""),
                                          C.byref(p), ob.ctypes.data, ob.size, oo.ctypes.data, of.ctypes.data, C.byref(need), C.byref(info))
            if rc != 2:
                break
            ob = np.empty(int(need.value) + 64, dtype=np.uint8)
        if rc == 1:
            raise HostFlattenerNeeded(load().cbh_last_error().decode("utf-8", "replace"))
        _check(rc)
        raw = ob[:int(oo[n])].tobytes()
        return [raw[int(oo[i]):int(oo[i + 1])] for i in range(n)], of[:n].copy()

    def wire_check_pb_submit(self, data, offsets, out, now_ns=0, flags=0, default_policy_version="default", default_scope="", device_index=0):
        """``cbh_wire_check_pb_submit``: the same call on a worker of the library.  ``out`` = (bytes uint8[cap], offsets uint64[n + 1],
        flags uint8[n]) - the caller's buffers, untouched until ``wire_check_pb_collect``.  -> a ticket (keeps the arrays alive)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        ob, oo, of = out
        p = CParams(now_ns, flags, 0)
        h = C.c_void_p()
        _check(load().cbh_wire_check_pb_submit(self.h, device_index, data.ctypes.data if data.size else None, offsets.ctypes.data, n,
                                               default_policy_version.encode(), default_scope.encode(), None, 0, C.byref(p),
                                               ob.ctypes.data, ob.size, oo.ctypes.data, of.ctypes.data, C.byref(h)))
        return (h, data, offsets, out, n)

    def wire_check_pb_collect(self, ticket):
        """``cbh_wire_check_pb_collect``: -> ([serialized CheckOutput], flags uint8[n]) of the submitted call."""
        h, _data, _offsets, (ob, oo, of), n = ticket
        info, need = CWireInfo(), C.c_size_t()
        rc = load().cbh_wire_check_pb_collect(self.h, h, C.byref(need), C.byref(info))
        if rc == 1:
            raise HostFlattenerNeeded(load().cbh_last_error().decode("utf-8", "replace"))
        if rc == 2:
            raise HipEngineError("cbh_wire_check_pb_collect: the output buffer is too small (%d bytes needed)" % need.value)
        _check(rc)
        raw = ob[:int(oo[n])].tobytes()
        return [raw[int(oo[i]):int(oo[i + 1])] for i in range(n)], of[:n].copy()

    def wire_check_requests_pb(self, requests, aux=None, now_ns=0, flags=0, default_policy_version="default", default_scope="", device_index=0,
                               globals_pb=b"", trail=False):
        """``cbh_wire_check_requests_pb``: serialized ``CheckResourcesRequest``s in (``requests``: [bytes]; ``aux``: per request the
        serialized engine ``AuxData`` or None), per request the serialized ``CheckOutput``s of its resource entries out.
        -> ([[bytes] per request], flags uint8[n_inputs], include_meta bool[n_requests]); with ``trail``
        (``cbh_wire_check_requests_trail_pb``) a fourth value: uint32[n_requests][words], the policies every request went through
        (``policy_keys`` names the bits)."""
        from .wire import pack_messages
        data, offsets = pack_messages(list(requests))
        nr = len(offsets) - 1
        a_data = a_off = None
        if aux is not None and any(aux):
            a_data, a_off = pack_messages([x or b"" for x in aux])
        first = np.zeros(nr + 1, dtype=np.uint32)
        rflags = np.zeros(max(nr, 1), dtype=np.uint8)
        p = CParams(now_ns, flags, 0)
        info, need = CWireInfo(), C.c_size_t()
        cap, n_cap = 4096, 8 * max(nr, 1)
        words = (int(load().cbh_table_num_policies(self.h)) + 31) // 32 if trail else 0
        ep = np.zeros((max(nr, 1), max(words, 1)), dtype=np.uint32)
        fn = load().cbh_wire_check_requests_trail_pb if trail else load().cbh_wire_check_requests_pb
        for _ in range(3):
            ob = np.empty(cap, dtype=np.uint8)
            oo, of = np.zeros(n_cap + 1, dtype=np.uint64), np.zeros(n_cap + 1, dtype=np.uint8)   # (the inputs are known after the call: grown on demand)
            rc = fn(self.h, device_index, data.ctypes.data if data.size else None, offsets.ctypes.data, nr,
                    a_data.ctypes.data if a_data is not None and a_data.size else (np.zeros(1, np.uint8).ctypes.data if a_off is not None else None),
                    a_off.ctypes.data if a_off is not None else None,
                    default_policy_version.encode(), default_scope.encode(), globals_pb or None, len(globals_pb or b""),
                    C.byref(p), first.ctypes.data, rflags.ctypes.data, ob.ctypes.data, ob.size, oo.ctypes.data, of.ctypes.data,
                    n_cap, C.byref(need), C.byref(info), *([ep.ctypes.data] if trail else []))
            if rc != 2:
                break
            cap, n_cap = max(cap, int(need.value) + 64), max(n_cap, int(info.n_requests))
        if rc == 1:
            raise HostFlattenerNeeded(load().cbh_last_error().decode("utf-8", "replace"))
        _check(rc)
        n = int(first[nr])
        raw = ob[:int(oo[n])].tobytes()
        outs = [raw[int(oo[i]):int(oo[i + 1])] for i in range(n)]
        res = [outs[int(first[r]):int(first[r + 1])] for r in range(nr)], of[:n].copy(), (rflags[:nr] & 1).astype(bool)
        return res + (ep[:nr, :words],) if trail else res

    def wire_spans(self, dbatch):
        """``cbh_wire_spans_download`` -> (in_span uint32[n][12], act_span uint32[n_tuples][2], act_off uint32[n + 1])"""
        n, T = dbatch.n_requests, dbatch.n_tuples
        in_span = np.zeros((max(n, 1), 12), dtype=np.uint32)
        act_span = np.zeros((max(T, 1), 2), dtype=np.uint32)
        act_off = np.zeros(n + 1, dtype=np.uint32)
        _check(load().cbh_wire_spans_download(self.h, dbatch.h, in_span.ctypes.data, act_span.ctypes.data, act_off.ctypes.data))
        return in_span[:n], act_span[:T], act_off

    def wire_outputs(self, dbatch, cap=None):
        """``cbh_wire_outputs``: the serialized CheckOutputs of a device-flattened batch, written by the GPU (after ``launch``)
        -> ([bytes], flags uint8[n])"""
        n = dbatch.n_requests
        cap = int(cap if cap is not None else 96 * max(n, 1))
        off = np.zeros(n + 1, dtype=np.uint64)
        flags = np.zeros(max(n, 1), dtype=np.uint8)
        while True:
            buf = np.zeros(max(cap, 1), dtype=np.uint8)
            need = C.c_size_t()
            rc = load().cbh_wire_outputs(self.h, dbatch.h, buf.ctypes.data, cap, off.ctypes.data, flags.ctypes.data, C.byref(need))
            if rc == 2:
                cap = int(need.value)
                continue
            _check(rc)
            raw = buf.tobytes()
            return [raw[int(off[i]):int(off[i + 1])] for i in range(n)], flags[:n]

    def launch(self, dbatch, now_ns=0, flags=0):
        p = CParams(now_ns, flags, 0)
        _check(load().cbh_check_resident(self.h, dbatch.h, C.byref(p)))

    def set_trail(self, dbatch, groups=None, n_groups=1):
        """``cbh_batch_set_trail``: the resident batch keeps an audit trail from now on (launch with ``F_WANT_EFFECTIVE_POLICIES``);
        ``groups``: the group of every request in the batch's device order, None = one group.  Clears the masks."""
        grp = None if groups is None else np.ascontiguousarray(groups, dtype=np.uint32)
        _check(load().cbh_batch_set_trail(self.h, dbatch.h, grp.ctypes.data if grp is not None else None, n_groups))
        dbatch.trail_groups = max(int(n_groups), 1)

    def trail(self, dbatch):
        """``cbh_trail_download`` -> uint32[n_groups][words]"""
        words = (int(load().cbh_table_num_policies(self.h)) + 31) // 32
        masks = np.zeros((dbatch.trail_groups, max(words, 1)), dtype=np.uint32)
        _check(load().cbh_trail_download(self.h, dbatch.h, masks.ctypes.data))
        return masks[:, :words]

    def launch_many(self, dbatches, now_ns=0, flags=0):
        """``cbh_check_resident_many``: one sweep over resident batches (a prepared handle array is cached per list)."""
        key = id(dbatches)
        arr = getattr(self, "_sweep", (None, None))
        if arr[0] != key or len(arr[1]) != len(dbatches):
            arr = (key, (C.c_void_p * len(dbatches))(*[db.h for db in dbatches]))
            self._sweep = arr
        p = CParams(now_ns, flags, 0)
        _check(load().cbh_check_resident_many(self.h, arr[1], len(dbatches), C.byref(p)))

    def set_resident_streams(self, n):
        _check(load().cbh_table_set_resident_streams(self.h, int(n)))

    @property
    def resident_streams(self):
        return int(load().cbh_table_resident_streams(self.h))

    def synchronize(self):
        _check(load().cbh_synchronize(self.h))

    def plan(self, dbatch, flags=0):
        """The kernels cbh_check_resident launches for this batch under `flags` (cbh_plan_describe)."""
        lib = load()
        lib.cbh_plan_describe.restype = C.c_char_p
        p = CParams(0, flags, 0)
        return lib.cbh_plan_describe(self.h, dbatch.h, C.byref(p)).decode()

    def kernel_time_ms(self):
        a, b = C.c_float(), C.c_float()
        _check(load().cbh_kernel_time_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def download(self, dbatch, want=("policy", "scope", "status", "edr")):
        res = Result(dbatch.n_tuples, dbatch.n_requests, want)
        _check(load().cbh_result_download(self.h, dbatch.h, C.byref(res.c)))
        return res.to_input_order(dbatch.order)


class DeviceBatch:
    def __init__(self, table, h, n_tuples, n_requests, batch=None):
        self.table, self.h, self.n_tuples, self.n_requests = table, h, n_tuples, n_requests

        class _Order:   # what is needed to map device order back to input order, without the big arrays
            tuple_perm = getattr(batch, "tuple_perm", None)
            req_perm = getattr(batch, "req_perm", None)
            vreq_input = getattr(batch, "vreq_input", None)
        self.order = _Order

    def close(self):
        if self.h:
            load().cbh_batch_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Result:
    def __init__(self, n_tuples, n_requests, want, pinned=False):
        """``pinned``: all five arrays in one page-locked result slab (cbh_result_bind_slab) - the wanted ones come
        back in one DMA each run of neighbours; otherwise ordinary numpy arrays."""
        self.effect = self.policy = self.scope = self.status = self.edr = None
        if pinned:
            lib = load()
            slab = pinned_empty(max(1, lib.cbh_result_slab_bytes(n_tuples, n_requests)), np.uint8)
            c = CResult()
            lib.cbh_result_bind_slab(C.byref(c), slab.ctypes.data_as(C.c_void_p), n_tuples, n_requests)
            self.slab = slab
            mk = lambda n, dt, addr: _slab_view(slab, addr, (n,), dt)   # noqa: E731
        else:
            c = CResult(0, 0, 0, 0, 0)
            mk = lambda n, dt, addr: np.zeros(n, dtype=dt)   # noqa: E731
        self.effect = mk(n_tuples, np.uint8, c.effect)
        if "policy" in want:
            self.policy = mk(n_tuples, np.uint32, c.policy)
        if "scope" in want:
            self.scope = mk(n_tuples, np.uint32, c.scope)
        if "status" in want:
            self.status = mk(n_tuples, np.uint8, c.status)
        if "edr" in want:
            self.edr = mk(n_requests, np.uint64, c.edr_mask)
        self.c = CResult(_ptr(self.effect) or 0, _ptr(self.policy), _ptr(self.scope), _ptr(self.status),
                         _ptr(self.edr))

    def to_input_order(self, batch):
        """Undo the flattener's routing sort / request splitting: per-tuple arrays back in input
        tuple order, ``edr`` indexed by input request."""
        tp = getattr(batch, "tuple_perm", None)
        if tp is not None:
            for name in ("effect", "policy", "scope", "status"):
                a = getattr(self, name)
                if a is not None:
                    out = np.empty_like(a)
                    out[tp] = a
                    setattr(self, name, out)
        if self.edr is not None:
            rp = getattr(batch, "req_perm", None)
            vi = getattr(batch, "vreq_input", None)
            edr = self.edr
            if rp is not None:
                e2 = np.empty_like(edr)
                e2[rp] = edr
                edr = e2
            if vi is not None and edr.size:
                out = np.zeros(int(vi.max()) + 1 if vi.size else 0, dtype=np.uint64)
                np.bitwise_or.at(out, vi, edr)
                edr = out
            self.edr = edr
        return self
