"""TEST INFRASTRUCTURE ONLY: ctypes driver of oracle/ccheck.cpp (the scalar C++ restatement of
check.go:97-460 over the lowered table image).  Used by tests/ for full-size parity and by
bench.py's cpu_baseline leg; the product never imports it.

The ctypes structure definitions of the C ABI (``cbh_batch`` / ``cbh_params`` / ``cbh_result``)
are taken from cerbos_amd.capi: they describe include/cerbos_hip.h, they are not evaluator code."""
import ctypes as C
import os
import subprocess

from cerbos_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "ccheck.cpp")
LIB = os.path.join(HERE, "libccheck.so")
_DEPS = [SRC, os.path.join(ROOT, "cerbos_amd", "csrc", "cbh_blob.h"), os.path.join(ROOT, "include", "cerbos_hip.h")]

_lib = None


class Unsupported(Exception):
    """The table uses features outside the C++ restatement (ccheck_run returned 1; none today)."""


def build(force=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _DEPS):
        return LIB
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cerbos_amd", "csrc"),
                           SRC, "-o", LIB])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.ccheck_run.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(capi.CBatch), C.POINTER(capi.CParams),
                                    C.POINTER(capi.CResult), C.c_int]
        _lib.ccheck_run.restype = C.c_int
    return _lib


class Prepared:
    """A batch marshalled once, so that bench.py can time ``run`` alone."""

    def __init__(self, lt, batch):
        self.lt, self.batch = lt, batch
        self.cb = capi.make_cbatch(batch, len(lt.columns))
        self.buf = C.create_string_buffer(lt.blob, len(lt.blob))

    def run(self, now_ns=0, flags=0, threads=1, want=("policy", "scope", "status", "edr")):
        res = capi.Result(self.batch.n_tuples, self.batch.n_requests, want)
        p = capi.CParams(now_ns, flags, 0)
        rc = lib().ccheck_run(C.cast(self.buf, C.c_void_p), len(self.lt.blob), C.byref(self.cb), C.byref(p),
                              C.byref(res.c), threads)
        if rc == 1:
            raise Unsupported("table outside the C++ restatement")
        if rc != 0:
            raise RuntimeError("ccheck_run failed: bad table image")
        return res


def check(lt, batch, now_ns=0, flags=0, threads=1):
    """Results in INPUT order (like capi.Table.check)."""
    return Prepared(lt, batch).run(now_ns, flags, threads).to_input_order(batch)
