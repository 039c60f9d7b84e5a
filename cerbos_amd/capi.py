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

ABI_VERSION = 2
F_LENIENT_SCOPE_SEARCH = 1
F_STRICT_EVALUATION = 2
F_WANT_DERIVED_ROLES = 4

EFFECT_ALLOW, EFFECT_DENY = 1, 2
ST_OK, ST_CEL_ERROR, ST_UNSUPPORTED = 0, 1, 2
P_EMPTY, P_NO_MATCH, P_RESOURCE, P_PRINCIPAL, P_TABLE, P_NO_MATCH_SP = range(6)
NONE = 0xFFFFFFFF

EXPORTED_SYMBOLS = [
    "cbh_init", "cbh_shutdown", "cbh_last_error", "cbh_abi_version",
    "cbh_table_load", "cbh_table_release", "cbh_table_num_strings", "cbh_table_num_columns",
    "cbh_table_device_bytes", "cbh_table_device_ptr", "cbh_table_adopt_device_image",
    "cbh_check_batch", "cbh_batch_upload", "cbh_batch_release", "cbh_check_resident",
    "cbh_synchronize", "cbh_result_download", "cbh_kernel_time_ms",
]


class HipEngineError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32)]


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
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise HipEngineError(load().cbh_last_error().decode("utf-8", "replace"))


_inited_device = None


def init(device: int = 0):
    global _inited_device
    lib = load()
    if lib.cbh_abi_version() != ABI_VERSION:
        raise HipEngineError("libcerbos_hip.so ABI mismatch")
    cfg = Config(ABI_VERSION, device)
    _check(lib.cbh_init(C.byref(cfg)))
    _inited_device = device


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
    def check(self, batch, now_ns=0, flags=0, want=("policy", "scope", "status", "edr"), device_order=False):
        res = Result(batch.n_tuples, batch.n_requests, want)
        cb = make_cbatch(batch, self.num_columns)
        p = CParams(now_ns, flags, 0)
        _check(load().cbh_check_batch(self.h, C.byref(cb), C.byref(p), C.byref(res.c)))
        return res if device_order else res.to_input_order(batch)

    # ---- resident
    def upload(self, batch):
        cb = make_cbatch(batch, self.num_columns)
        h = C.c_void_p()
        _check(load().cbh_batch_upload(self.h, C.byref(cb), C.byref(h)))
        return DeviceBatch(self, h, batch.n_tuples, batch.n_requests, batch)

    def launch(self, dbatch, now_ns=0, flags=0):
        p = CParams(now_ns, flags, 0)
        _check(load().cbh_check_resident(self.h, dbatch.h, C.byref(p)))

    def synchronize(self):
        _check(load().cbh_synchronize(self.h))

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
    def __init__(self, n_tuples, n_requests, want):
        self.effect = np.zeros(n_tuples, dtype=np.uint8)
        self.policy = np.zeros(n_tuples, dtype=np.uint32) if "policy" in want else None
        self.scope = np.zeros(n_tuples, dtype=np.uint32) if "scope" in want else None
        self.status = np.zeros(n_tuples, dtype=np.uint8) if "status" in want else None
        self.edr = np.zeros(n_requests, dtype=np.uint64) if "edr" in want else None
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
