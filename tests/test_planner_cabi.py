"""PlanResources over the C ABI (include/cerbos_lower.h cbl_planner_*: what a Go server calls beside cbl_lower_ruletable_pb): the
reference's 116 query-planner plans as serialized PlanResourcesInputs in, serialized PlanResourcesOutputs out - the planner opened from
the serialized runtimev1.RuleTable - against the goldens and against the planner called directly."""
import ctypes as C
import json
import os

import pytest

import __graft_entry__
from cerbos_amd import wire
from cerbos_amd.plan import Planner
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from cerbos_amd.ruletable.proto import encode_rule_table
from test_planner_golden import AUX, DATA, NOW, canon

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cerbos_amd", "libcerbos_lower.so")


@pytest.fixture(scope="module")
def lib():
    __graft_entry__.build_lower()
    so = C.CDLL(LIB)
    so.cbl_planner_open.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
    so.cbl_planner_open.restype = C.c_int
    so.cbl_planner_plan_pb.argtypes = [C.c_uint64, C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
    so.cbl_planner_plan_pb.restype = C.c_int
    so.cbl_planner_close.argtypes = [C.c_uint64]
    so.cbl_free.argtypes = [C.c_void_p]
    return so


def _plan(so, h, inp, params):
    pb = wire.encode_plan_resources_input(inp)
    out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
    st = so.cbl_planner_plan_pb(h, pb, len(pb), json.dumps(params).encode(), C.byref(out), C.byref(n), C.byref(err))
    try:
        return st, (wire.decode_plan_resources_output(C.string_at(out, n.value)) if st == 0 else C.string_at(err).decode())
    finally:
        so.cbl_free(out)
        so.cbl_free(err)


def test_the_reference_s_plans_through_bytes(lib):
    rt = rule_table_from_policies(policies_from_docs(DATA["policies"]))
    pb = encode_rule_table(rt)
    h, err = C.c_uint64(), C.c_void_p()
    assert lib.cbl_planner_open(pb, len(pb), C.byref(h), C.byref(err)) == 0, C.string_at(err)
    direct = Planner(rt)
    n = 0
    try:
        for suite in DATA["suites"]:
            for test in suite["tests"]:
                if test["wantErr"]:
                    continue
                for lenient in ([False, True] if suite["lenient"] is None else [suite["lenient"]]):
                    inp = {"requestId": "requestId", "principal": suite["principal"], "resource": test["resource"], "actions": test["actions"], "auxData": AUX}
                    st, out = _plan(lib, h.value, inp, {"globals": {"environment": "test"}, "lenientScopeSearch": lenient, "nowNs": NOW})
                    assert st == 0, out
                    assert out["filter"]["kind"] == test["want"].get("kind")
                    assert canon(out["filter"].get("condition")) == canon(test["want"].get("condition")), (suite["name"], test["actions"])
                    want = direct.plan(inp, globals_={"environment": "test"}, lenient_scope_search=lenient, now_ns=NOW)
                    assert canon(out["filter"].get("condition")) == canon(want["filter"].get("condition"))
                    assert out["filterDebug"] == want["filterDebug"] and out["matchedScopes"] == want["matchedScopes"]
                    assert out["actions"] == test["actions"] and out["kind"] == test["resource"].get("kind", "") and out["requestId"] == "requestId"
                    n += 1
        assert n > 200
        # the deprecated single `action` (PlanResourcesInput field 2) comes back as it came: Action set, Actions empty
        # (planner.go:113-125 MkPlanResourcesOutput copies both verbatim) - and plans as [action]
        suite = DATA["suites"][0]
        test = next(t for t in suite["tests"] if not t["wantErr"] and len(t["actions"]) == 1)
        one = {"requestId": "r", "principal": suite["principal"], "resource": test["resource"], "action": test["actions"][0], "auxData": AUX}
        many = dict(one, actions=test["actions"])
        del many["action"]
        params = {"globals": {"environment": "test"}, "lenientScopeSearch": bool(suite["lenient"]), "nowNs": NOW}
        (st1, o1), (st2, o2) = _plan(lib, h.value, one, params), _plan(lib, h.value, many, params)
        assert st1 == 0 and st2 == 0
        assert o1["action"] == test["actions"][0] and o1["actions"] == []
        assert o2["action"] == "" and o2["actions"] == test["actions"]
        assert o1["filter"] == o2["filter"] and o1["matchedScopes"] == o2["matchedScopes"]
        st, msg = _plan(lib, 987654, {"principal": {"id": "x", "roles": ["a"]}, "resource": {"kind": "k"}, "actions": ["a"]}, {})
        assert st == 3 and "handle" in msg
    finally:
        lib.cbl_planner_close(h.value)
    st, msg = _plan(lib, h.value, {"principal": {"id": "x", "roles": ["a"]}, "resource": {"kind": "k"}, "actions": ["a"]}, {})
    assert st == 3       # closed


def test_bytes_that_are_no_rule_table(lib):
    h, err = C.c_uint64(), C.c_void_p()
    junk = b"\x0a\xff\xff\xff\xff\x0f nope"
    assert lib.cbl_planner_open(junk, len(junk), C.byref(h), C.byref(err)) == 3
    assert b"bad input" in C.string_at(err)
    lib.cbl_free(err)
