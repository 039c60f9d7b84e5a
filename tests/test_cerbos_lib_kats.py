"""TestCerbosLib (internal/conditions/cerbos_lib_test.go:26-193), the reference's Go-coded known-answer tests of its CEL
library - 143 closed expressions that must be true, one that must fail - mined into tests/golden/cerbos_lib_kats.json
(tools/make_golden_cerbos_lib.py).

* the oracle (oracle/celeval.py, oracle/crosspath.py): the answer for every expression;
* the device path (kernel source on the host simulator; GPU tier: the kernel): each expression as the condition of an ALLOW
  rule - ALLOW where the KAT says true, a CEL error where it says error; every one of them decided (none flagged
  UNSUPPORTED), never a wrong answer."""
import pytest

from cerbos_amd import capi
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json
from oracle import celeval

CASES = load_json("cerbos_lib_kats.json")["cases"]
NOW = 1_700_000_000_000_000_000
API = "api.cerbos.dev/v1"
# function families oracle/celeval.py does not restate: none (the file-path helpers: oracle/crosspath.py, SPIFFE: celeval.py)
ORACLE_GAPS = ()


def _oracle_implements(expr):
    return not any(g in expr for g in ORACLE_GAPS)


@pytest.mark.parametrize("case", [c for c in CASES if _oracle_implements(c["expr"])], ids=lambda c: c["expr"][:60])
def test_oracle_answers_as_the_reference_asserts(case):
    env = celeval.Env({}, NOW)
    if case["wantErr"]:
        with pytest.raises(celeval.CelError):
            celeval.evaluate(case["expr"], env)
    else:
        assert celeval.evaluate(case["expr"], env) is True, case["expr"]


def test_oracle_coverage_of_the_library_kats():
    done = sum(_oracle_implements(c["expr"]) for c in CASES)
    assert done == len(CASES) == 143, done


def _device(make, close):
    """All expressions in ONE table: rule k allows action "k" under expression k."""
    rules = [{"actions": ["a%d" % k], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": c["expr"]}}}
             for k, c in enumerate(CASES)]
    rt = rule_table_from_policies(policies_from_docs([{"apiVersion": API, "resourcePolicy": {"resource": "kat", "version": "default", "rules": rules}}]))
    lt = lower_rule_table(rt)
    ev = make(lt)
    decided = wrong = 0
    try:
        for lo in range(0, len(CASES), 48):
            ks = list(range(lo, min(lo + 48, len(CASES))))
            inp = {"requestId": "kat", "principal": {"id": "p", "roles": ["user"]}, "resource": {"kind": "kat", "id": "r"},
                   "actions": ["a%d" % k for k in ks]}
            batch = Flattener(lt).flatten([inp])
            res = ev.table.check(batch, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
            for j, k in enumerate(ks):
                st, eff = int(res.status[j]), int(res.effect[j])
                if st == capi.ST_UNSUPPORTED:
                    continue
                decided += 1
                if CASES[k]["wantErr"]:
                    ok = st == capi.ST_CEL_ERROR and eff != 1
                else:
                    ok = eff == 1 and st == capi.ST_OK
                if not ok:
                    wrong += 1
                    print("WRONG", CASES[k], st, eff)
    finally:
        if close:
            ev.close()
    return decided, wrong


def test_kernel_source_never_answers_a_library_kat_wrongly():
    from test_hostsim_golden import HostSimEvaluator
    decided, wrong = _device(lambda lt: HostSimEvaluator(lt, Conf()), False)
    assert wrong == 0
    assert decided == len(CASES), decided


@pytest.mark.gpu
def test_gpu_never_answers_a_library_kat_wrongly():
    decided, wrong = _device(lambda lt: HipEvaluator(lt, Conf()), True)
    assert wrong == 0 and decided == len(CASES), (decided, wrong)
