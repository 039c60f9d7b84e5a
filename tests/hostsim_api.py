"""TEST INFRASTRUCTURE ONLY: ctypes driver for tests/hostsim/libcbh_hostsim.so - the device
source of the decision kernel compiled for the host (see tests/hostsim/hostsim.cpp).
Lets the CPU tier check lowering + bytecode + kernel logic without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np

from cerbos_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "hostsim", "hostsim.cpp")
LIB = os.path.join(HERE, "hostsim", "libcbh_hostsim.so")
_DEPS = [SRC] + [os.path.join(ROOT, "cerbos_amd", "csrc", f) for f in
                 ("cbh_kernels.h", "cbh_check_wave.h", "cbh_check_flat.h", "cbh_check_walk2.h", "cbh_interp.h", "cbh_vm.h", "cbh_blob.h", "cbh_image.h", "cbh_wire.h", "cbh_wire_host.h")] + [os.path.join(ROOT, "include", "cerbos_hip.h")]

_lib = None


def build():
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _DEPS):
        return
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           SRC, "-o", LIB])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.hostsim_check.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(capi.CBatch), C.POINTER(capi.CParams),
                                       C.POINTER(capi.CResult), C.c_void_p]
        _lib.hostsim_check.restype = C.c_int
        _lib.hostsim_trace.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(capi.CBatch), C.POINTER(capi.CParams),
                                       C.POINTER(capi.CResult), C.c_void_p, C.POINTER(capi.CTrace)]
        _lib.hostsim_trace.restype = C.c_int
        _lib.hostsim_last_error.restype = C.c_char_p
        _lib.hostsim_last_kind.restype = C.c_int
        _lib.hostsim_last_walk_wide.restype = C.c_int
    return _lib


def batch_gbits(lt, batch):
    """[3][n_strings] glob bits of batch-local strings via the host simulation of the device NFA."""
    g = np.zeros((3, max(batch.n_strings, 1)), dtype=np.uint64)
    off, data, flags = batch.str_off, batch.str_bytes.tobytes(), batch.str_flags
    for i in range(batch.n_strings):
        s = data[off[i]:off[i + 1]]
        for d in range(3):
            if flags[i] & (1 << d) and lt.nfas[d].patterns:
                g[d, i] = lt.nfas[d].match_bits(s)
    return g[:, :batch.n_strings].copy() if batch.n_strings else g


def check(lt, batch, now_ns=0, flags=0, device_order=False):
    res = capi.Result(batch.n_tuples, batch.n_requests, ("policy", "scope", "status", "edr"))
    cb = capi.make_cbatch(batch, len(lt.columns))
    p = capi.CParams(now_ns, flags, 0)
    g = batch_gbits(lt, batch)
    buf = C.create_string_buffer(lt.blob, len(lt.blob))
    rc = lib().hostsim_check(C.cast(buf, C.c_void_p), len(lt.blob), C.byref(cb), C.byref(p), C.byref(res.c),
                             g.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(lib().hostsim_last_error().decode())
    return res if device_order else res.to_input_order(batch)


def check_trail(lt, batch, groups=None, n_groups=1, now_ns=0, flags=0):
    """cbh_check_batch_trail on the simulator -> (Result in device order, masks uint32[n_groups][words]); ``groups``: group of every
    request of ``batch`` (device order), or None = one group."""
    res = capi.Result(batch.n_tuples, batch.n_requests, ("policy", "scope", "status", "edr"))
    cb = capi.make_cbatch(batch, len(lt.columns))
    p = capi.CParams(now_ns, flags, 0)
    g = batch_gbits(lt, batch)
    buf = C.create_string_buffer(lt.blob, len(lt.blob))
    n_pol = len(lt.policy_keys)
    words = (n_pol + 31) // 32
    masks = np.zeros((max(n_groups, 1), max(words, 1)), dtype=np.uint32)
    grp = None if groups is None else np.ascontiguousarray(groups, dtype=np.uint32)
    lib().hostsim_check_trail.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    rc = lib().hostsim_check_trail(C.cast(buf, C.c_void_p), len(lt.blob), C.byref(cb), C.byref(p), C.byref(res.c), g.ctypes.data_as(C.c_void_p),
                                   grp.ctypes.data if grp is not None else None, n_groups, n_pol, masks.ctypes.data)
    if rc != 0:
        raise RuntimeError(lib().hostsim_last_error().decode())
    return res, masks[:, :words]


def last_pre_split():
    """Did the last batch run the walk's pre-pass as collector + interpreter (CBH_PRE_SPLIT=1)?"""
    return bool(lib().hostsim_last_pre_split())


def last_kind():
    """Which kernel family decided the last batch: 0 the general walk, 1 a flat kernel, 2 cbh_walk2_kernel."""
    return lib().hostsim_last_kind()


def last_walk_wide():
    """Which wider walks the last batch launched: bit 0 cbh_walk2_wide_kernel (requests with five to eight roles), bit 1
    cbh_walk2_awide_kernel (nine to sixteen actions)."""
    return int(lib().hostsim_last_walk_wide())


def trace(lt, batch, now_ns=0, flags=0, capacity=None):
    """The trace pass (cbh_trace_batch) on the simulator -> (Result in device order, records uint32[n][8])."""
    cap = capacity or max(256, 4 * batch.n_tuples)
    cb = capi.make_cbatch(batch, len(lt.columns))
    p = capi.CParams(now_ns, flags, 0)
    g = batch_gbits(lt, batch)
    buf = C.create_string_buffer(lt.blob, len(lt.blob))
    while True:
        res = capi.Result(batch.n_tuples, batch.n_requests, ("policy", "scope", "status", "edr"))
        rec = np.zeros((cap, capi.TRACE_RECORD_WORDS), dtype=np.uint32)
        tr = capi.CTrace(rec.ctypes.data, cap, 0)
        rc = lib().hostsim_trace(C.cast(buf, C.c_void_p), len(lt.blob), C.byref(cb), C.byref(p), C.byref(res.c),
                                 g.ctypes.data_as(C.c_void_p), C.byref(tr))
        if rc != 0:
            raise RuntimeError(lib().hostsim_last_error().decode())
        if tr.count <= cap:
            return res, rec[:tr.count]
        cap = int(tr.count) + 64
