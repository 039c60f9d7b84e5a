"""Arithmetic and ordering the coverage table of the round (profiles/r04_source_coverage_gpu_tier_on_simulator.txt) showed no test
reached ON REQUEST VALUES (constants fold at lowering): uint arithmetic with its overflow / underflow / zero-divisor errors, uint
against int and double, bool ordering, int64 subtraction overflow and modulus errors, duration sums - device (the kernel simulation in the
CPU tier, the kernel on the GPU tier) against the oracle, every (expression, request) pair; the expected values of a few are
written out by hand from cel-go's overloads (common/types/uint.go, int.go, double.go compare*; overflow.go)."""
import pytest

from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
CONDS = {
    "uadd": "uint(R.attr.n) + 2u == 7u",
    "usub": "uint(R.attr.n) - 9u == 0u",                      # underflow below 9: an error, the rule does not fire
    "umul": "uint(R.attr.n) * 3u > 10u",
    "udiv": "uint(R.attr.n) / uint(R.attr.z) == 1u",          # z = 0: division by zero
    "umod": "uint(R.attr.n) % 4u == 1u",
    "umodz": "uint(R.attr.n) % uint(R.attr.z) == 0u",
    "ulit": "uint(R.attr.n) < 6u",
    "uint_int": "uint(R.attr.n) < int(R.attr.m)",             # cross-type ordering (cel-go: compareUintInt)
    "int_uint": "int(R.attr.m) <= uint(R.attr.n)",
    "uint_dbl": "uint(R.attr.n) < R.attr.f",                  # compareUintDouble
    "dbl_uint": "R.attr.f >= uint(R.attr.n)",
    "uint_eq_dbl": "uint(R.attr.n) == R.attr.f",
    "boolord": "R.attr.b1 < R.attr.b2",
    "isub": "int(R.attr.m) - int(R.attr.big) < 0",            # big = -2^63: overflow for m >= 0
    "imod": "int(R.attr.n) % int(R.attr.z) == 0",             # modulus by zero
    "iorder": "int(R.attr.n) < int(R.attr.m)",
    "dursum": 'duration(R.attr.d1) + duration(R.attr.d2) > duration("1h")',
    "strord": "R.attr.s1 < R.attr.s2",
    "strin": 'R.attr.s1 in ["a", "b", "c"]',
}
ATTRS = [
    {"n": 5, "z": 5, "m": 7, "f": 5.5, "b1": False, "b2": True, "big": -9223372036854775808, "d1": "40m", "d2": "30m", "s1": "a", "s2": "ab"},
    {"n": 9, "z": 0, "m": -3, "f": 9.0, "b1": True, "b2": True, "big": 5, "d1": "10m", "d2": "20m", "s1": "b", "s2": "a"},
    {"n": 1, "z": 1, "m": 1, "f": 0.5, "b1": True, "b2": False, "big": -1, "d1": "59m", "d2": "1m", "s1": "zz", "s2": "zz"},
    {"n": 13, "z": 3, "m": 13, "f": -2.0, "b1": False, "b2": False, "big": 0, "d1": "2h", "d2": "-90m", "s1": "", "s2": "a"},
    {"n": 4, "z": 4, "m": 100, "f": 1e30, "b1": False, "b2": True, "big": -9223372036854775808, "d1": "1h", "d2": "1ns", "s1": "c", "s2": "C"},
]
# by hand, for the first request (n=5, z=5, m=7, f=5.5, big=-2^63): uadd 5+2==7; usub 5-9 underflows (error -> no ALLOW); umul 15>10;
# udiv 1==1; umod 5%4==1; umodz 5%5==0; ulit 5<6; uint_int 5<7; int_uint 7<=5 false; uint_dbl 5<5.5; dbl_uint 5.5>=5; uint_eq_dbl false;
# boolord false<true; isub 7-(-2^63) overflows (error); imod 5%5==0; iorder 5<7; dursum 70m>1h; strord "a"<"ab"; strin true
FIRST = {"uadd": True, "usub": False, "umul": True, "udiv": True, "umod": True, "umodz": True, "ulit": True, "uint_int": True, "int_uint": False,
         "uint_dbl": True, "dbl_uint": True, "uint_eq_dbl": False, "boolord": True, "isub": False, "imod": True, "iorder": True, "dursum": True,
         "strord": True, "strin": True}


def _run(make_evaluator, close):
    docs = [{"apiVersion": API, "resourcePolicy": {"resource": "nums", "version": "default", "rules": [
        {"actions": [n], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": e}}} for n, e in CONDS.items()]}}]
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    inputs = [{"requestId": "q%d" % i, "actions": list(CONDS), "resource": {"kind": "nums", "id": "r", "attr": a},
               "principal": {"id": "p", "roles": ["user"], "attr": {}}} for i, a in enumerate(ATTRS)]
    ev = make_evaluator(lt)
    try:
        outs, bad = ev.check(inputs, now_ns=0, allow_unsupported=True)
    finally:
        if close:
            ev.close()
    orc = RuleTableOracle(rt)
    decided = 0
    for k, (inp, have) in enumerate(zip(inputs, outs)):
        want = orc.check(inp, EvalParams(now_ns=0))
        if k == 0:
            assert {n: want["actions"][n]["effect"] == "EFFECT_ALLOW" for n in CONDS} == FIRST
        if k in bad:
            continue          # (flagged for the CPU path: loud, never a different answer)
        decided += 1
        for name in CONDS:
            assert have["actions"][name]["effect"] == want["actions"][name]["effect"], (name, inp["resource"]["attr"])
    assert decided >= 4, (decided, lt.unsupported)


@pytest.mark.parametrize("env", [{}, {"CBH_NO_WALK2": "1"}], ids=["cbh_walk2_kernel and its pre-pass", "the general walk"])
def test_on_the_kernel_simulation(env, monkeypatch):
    """... through the interpreter as the walk's pre-pass calls it and as the general walk does (these conditions are nobody's leaves)."""
    import hostsim_api
    from test_hostsim_golden import HostSimEvaluator
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _run(lambda lt: HostSimEvaluator(lt, Conf()), False)
    assert hostsim_api.last_kind() == (0 if env else 2)


@pytest.mark.gpu
def test_on_the_gpu():
    _run(lambda lt: HipEvaluator(lt, Conf()), True)
