"""Pins oracle/ccheck.cpp (the scalar C++ restatement used for full-size parity and as the CPU
baseline) against oracle/check.py, which is itself pinned on the reference's golden fixtures
(tests/test_oracle_golden.py):

* the synthetic BASELINE configurations in every evaluation mode - effect, policy key, scope,
  effective derived roles of every tuple, and whether the request produced evaluation errors;
* the reference's own golden store (with and without its role policies) over the inputs of all golden
  engine cases, C5, and fuzzed stores with role policies / parent roles / principal policies.
"""
import json
import os

import numpy as np
import pytest

from cerbos_amd import capi, workloads
from cerbos_amd.engine import HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle import ccheck
from oracle.check import EvalParams, RuleTableOracle

NOW = 1_700_000_000_000_000_000
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CONFIGS = {
    "C1": (lambda: workloads.c1_policies(2), lambda n: workloads.c1_requests(n, n_sets=2)),
    "C2": (workloads.c2_policies, lambda n: workloads.c2_requests(n)),
    "C3": (workloads.c3_policies, lambda n: workloads.c3_requests(n)),
}


class _Decoder(HipEvaluator):
    """ids -> CheckOutput through the product's own assembly code, without a GPU table."""

    def __init__(self, lt):   # noqa: D401 - deliberately skips HipEvaluator.__init__ (needs a device)
        self.lt = lt


def _compare(rt, lt, inputs, batch, mode, threads=1, check_errors=True, globals_=None):
    flags = capi.F_WANT_DERIVED_ROLES
    flags |= capi.F_LENIENT_SCOPE_SEARCH if mode == "lenient" else 0
    flags |= capi.F_STRICT_EVALUATION if mode == "strict" else 0
    res = ccheck.check(lt, batch, NOW, flags, threads)
    outs, bad = _Decoder(lt).assemble(inputs, batch, res, "default", allow_unsupported=True)
    orc = RuleTableOracle(rt)
    params = EvalParams(globals_=globals_, now_ns=NOW, lenient_scope_search=mode == "lenient", strict_evaluation=mode == "strict")
    t = 0
    n_cmp = 0
    for r, (inp, got) in enumerate(zip(inputs, outs)):
        n_act = len(batch.actions_per_request[r])
        st = res.status[t:t + n_act]
        t += n_act
        if r in bad:
            continue
        want = orc.check(inp, params)
        for a, w in want["actions"].items():
            assert got["actions"][a] == {"effect": w["effect"], "policy": w["policy"], "scope": w.get("scope", "")}, (r, a, inp)
        assert sorted(got["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles", [])), (r, inp)
        if check_errors:
            assert bool((st == capi.ST_CEL_ERROR).any()) == bool(want.get("evaluationErrors")), (r, inp, want.get("evaluationErrors"))
        n_cmp += 1
    return n_cmp, len(bad)


@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("mode", ["default", "lenient", "strict"])
def test_ccheck_matches_python_oracle_on_configs(name, mode):
    pol_fn, req_fn = CONFIGS[name]
    rt = rule_table_from_policies(policies_from_docs(pol_fn()))
    lt = lower_rule_table(rt)
    cr = req_fn(500)
    inputs = cr.to_inputs()
    n_cmp, n_bad = _compare(rt, lt, inputs, Flattener(lt).flatten(inputs), mode)
    assert n_bad == 0 and n_cmp == len(inputs)


def test_ccheck_threads_agree():
    rt = rule_table_from_policies(policies_from_docs(workloads.c3_policies()))
    lt = lower_rule_table(rt)
    batch = workloads.c3_requests(20_000).to_batch(Flattener(lt))
    a = ccheck.check(lt, batch, NOW, 0, threads=1)
    b = ccheck.check(lt, batch, NOW, 0, threads=4)
    for f in ("effect", "policy", "scope", "status", "edr"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_ccheck_on_golden_store_without_role_policies():
    with open(os.path.join(GOLDEN, "store_policies.json")) as fh:
        docs = [d for d in json.load(fh) if "rolePolicy" not in d]
    with open(os.path.join(GOLDEN, "engine_cases.json")) as fh:
        cases = json.load(fh)
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    inputs = [inp for c in cases for inp in c["inputs"]]
    assert len(inputs) > 50
    total = skipped = 0
    for mode in ("default", "lenient", "strict"):
        # error presence is not compared here: the reference evaluates a policy's variables eagerly and
        # records their errors even when no condition reads them; the lowering inlines variables
        n_cmp, n_bad = _compare(rt, lt, inputs, Flattener(lt).flatten(inputs, sort=False), mode, check_errors=False)
        total += n_cmp
        skipped += n_bad
    # requests whose policies use general CEL programs are reported unsupported, never guessed
    assert total > skipped, (total, skipped)


def test_ccheck_on_golden_store_with_role_policies():
    """The whole golden store (role policies, parent roles, scope permissions) over every golden input."""
    with open(os.path.join(GOLDEN, "store_policies.json")) as fh:
        docs = json.load(fh)
    with open(os.path.join(GOLDEN, "engine_cases.json")) as fh:
        cases = json.load(fh)
    assert any("rolePolicy" in d for d in docs)
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt, {"environment": "test"})
    inputs = [inp for c in cases for inp in c["inputs"]]
    total = skipped = 0
    for mode in ("default", "lenient", "strict"):
        n_cmp, n_bad = _compare(rt, lt, inputs, Flattener(lt).flatten(inputs, sort=False), mode, check_errors=False,
                                globals_={"environment": "test"})
        total += n_cmp
        skipped += n_bad
    assert total > 2 * skipped, (total, skipped)


@pytest.mark.parametrize("mode", ["default", "lenient", "strict"])
def test_ccheck_matches_python_oracle_on_c5(mode):
    """C5: principal policies, role policies, action globs, nested CEL.  Requests whose path needs the
    general CEL interpreter are reported unsupported by ccheck (it evaluates leaf trees only)."""
    rt = rule_table_from_policies(policies_from_docs(workloads.c5_policies()))
    lt = lower_rule_table(rt)
    inputs = workloads.c5_requests(300).to_inputs()
    n_cmp, n_bad = _compare(rt, lt, inputs, Flattener(lt).flatten(inputs), mode)
    assert n_cmp > 50, (n_cmp, n_bad)


@pytest.mark.parametrize("seed", range(12))
def test_ccheck_matches_python_oracle_on_fuzzed_stores(seed):
    from cerbos_amd.lower.celc import LoweringError
    from test_fuzz_parity import _policies, _requests
    rng = np.random.default_rng(10_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    inputs = _requests(rng, 120)
    total = 0
    for mode in ("default", "lenient", "strict"):
        n_cmp, _ = _compare(rt, lt, inputs, Flattener(lt).flatten(inputs), mode, check_errors=False)
        total += n_cmp
    assert total > 100
