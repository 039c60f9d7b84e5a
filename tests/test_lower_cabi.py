"""libcerbos_lower.so (include/cerbos_lower.h): the lowering behind a C ABI for a host that is not Python.

* from a C program (tests/lower_host/lower_host.c stands for the Go server's cgo binding): serialized runtimev1.RuleTable in,
  an image out that is byte for byte the package's own lowering; three calls from three threads give the same bytes;
* from this process through ctypes (the interpreter is already running: the library joins it);
* per-call globals / no-trace flags, a table the lowering refuses, bytes that are no RuleTable."""
import ctypes as C
import json
import os
import subprocess

import pytest

import __graft_entry__
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from cerbos_amd.ruletable.proto import decode_rule_table, encode_rule_table
from helpers import store_rule_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cerbos_amd", "libcerbos_lower.so")
API = "api.cerbos.dev/v1"


@pytest.fixture(scope="module")
def lib():
    __graft_entry__.build_lower()
    so = C.CDLL(LIB)
    so.cbl_lower_ruletable_pb.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
    so.cbl_lower_ruletable_pb.restype = C.c_int
    so.cbl_last_stats_json.restype = C.c_void_p
    so.cbl_free.argtypes = [C.c_void_p]
    return so


def _call(so, pb, globals_json=None, flags=0):
    image, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
    st = so.cbl_lower_ruletable_pb(pb, len(pb), globals_json, flags, C.byref(image), C.byref(n), C.byref(err))
    try:
        return st, (C.string_at(image, n.value) if st == 0 else C.string_at(err).decode())
    finally:
        so.cbl_free(image)
        so.cbl_free(err)


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lower_host") / "lower_host")
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tests", "lower_host", "lower_host.c"), "-ldl", "-pthread"])
    return exe


def test_a_c_host_lowers_the_golden_store(lib, host, tmp_path):
    pb = encode_rule_table(store_rule_table())
    (tmp_path / "rt.pb").write_bytes(pb)
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "CERBOS_AMD_ROOT")}   # the library finds the package by its own place
    r = subprocess.run([host, LIB, str(tmp_path / "rt.pb"), str(tmp_path / "rt.img")], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = lower_rule_table(decode_rule_table(pb))
    assert (tmp_path / "rt.img").read_bytes() == bytes(want.blob)
    stats = json.loads(r.stdout)
    assert stats["unsupported"] == [list(x) for x in want.unsupported] and stats.keys() >= want.stats.keys()


def test_a_c_host_gets_the_refusal_and_the_flags(lib, host, tmp_path):
    (tmp_path / "junk.pb").write_bytes(b"\x0a\xff\xff\xff\xff\x0f not a rule table")
    r = subprocess.run([host, LIB, str(tmp_path / "junk.pb"), str(tmp_path / "x.img")], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "bad input" in r.stderr
    docs = [{"apiVersion": API, "resourcePolicy": {"resource": "doc", "version": "default", "rules": [
        {"actions": ["view"], "roles": ["user"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": "G.env == R.attr.env"}}}]}}]
    pb = encode_rule_table(rule_table_from_policies(policies_from_docs(docs)))
    (tmp_path / "g.pb").write_bytes(pb)
    r = subprocess.run([host, LIB, str(tmp_path / "g.pb"), str(tmp_path / "g1.img"), "1"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "g1.img").read_bytes() == bytes(lower_rule_table(decode_rule_table(pb), per_call_globals=True).blob)
    r = subprocess.run([host, LIB, str(tmp_path / "g.pb"), str(tmp_path / "g2.img"), "2", '{"env": "prod"}'], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "g2.img").read_bytes() == bytes(lower_rule_table(decode_rule_table(pb), {"env": "prod"}, trace=False).blob)
    r = subprocess.run([host, LIB, str(tmp_path / "g.pb"), str(tmp_path / "g3.img"), "0", "[1]"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "JSON object" in r.stderr


def test_inside_a_running_interpreter(lib):
    pb = encode_rule_table(store_rule_table())
    st, image = _call(lib, pb)
    assert st == 0 and image == bytes(lower_rule_table(decode_rule_table(pb)).blob)
    st, msg = _call(lib, b"\x08")
    assert st == 3 and "bad input" in msg
    assert lib.cbl_abi_version() == 1


def test_statistics_come_back_with_the_image(lib):
    """cbl_lower_ruletable_pb_stats: no thread-local in between (a goroutine may resume on another OS thread)"""
    lib.cbl_lower_ruletable_pb_stats.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.cbl_lower_ruletable_pb_stats.restype = C.c_int
    pb = encode_rule_table(store_rule_table())
    image, n, stats, err = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_void_p()
    assert lib.cbl_lower_ruletable_pb_stats(pb, len(pb), None, 0, C.byref(image), C.byref(n), C.byref(stats), C.byref(err)) == 0
    want = lower_rule_table(decode_rule_table(pb))
    assert C.string_at(image, n.value) == bytes(want.blob)
    got = json.loads(C.string_at(stats).decode())
    assert got["unsupported"] == [list(x) for x in want.unsupported] and got.keys() >= want.stats.keys()
    for p in (image, stats, err):
        lib.cbl_free(p)
