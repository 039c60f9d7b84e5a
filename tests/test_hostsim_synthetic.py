"""CPU tier: kernel logic (host-simulated device source) vs the oracle on the synthetic
BASELINE configs at sizes the oracle finishes in seconds, through BOTH ingest routes
(JSON-shaped CheckInputs -> Flattener, and ColumnarRequests -> vectorised SoA)."""
import numpy as np
import pytest

import hostsim_api
from cerbos_amd import capi, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle.check import EvalParams, RuleTableOracle

NOW = 1_700_000_000_000_000_000
EFF = {"EFFECT_ALLOW": capi.EFFECT_ALLOW, "EFFECT_DENY": capi.EFFECT_DENY}

CONFIGS = {
    "C1": (lambda: workloads.c1_policies(2), lambda n: workloads.c1_requests(n, n_sets=2)),
    "C2": (workloads.c2_policies, lambda n: workloads.c2_requests(n)),
    "C3": (workloads.c3_policies, lambda n: workloads.c3_requests(n)),
    # C4 at 1/10 of its table size here (the GPU tier runs the 1000-policy table); C5 whole
    "C4": (lambda: workloads.c4_policies(n_policies=99), lambda n: workloads.c4_requests(n, n_policies=99)),
    "C5": (workloads.c5_policies, lambda n: workloads.c5_requests(n)),
    # north_star's target set (100 policies / 10k rules, 100 rules per (kind, scope) bucket) and C5 with 5-8 roles per principal
    "T": (workloads.t_policies, lambda n: workloads.t_requests(n)),
    "C5W": (workloads.c5_policies, lambda n: workloads.c5w_requests(n)),
}


def oracle_effects(rt, inputs, **kw):
    orc = RuleTableOracle(rt)
    params = EvalParams(now_ns=NOW, **kw)
    eff = []
    for inp in inputs:
        out = orc.check(inp, params)
        eff.extend(EFF[out["actions"][a]["effect"]] for a in inp["actions"])
    return np.array(eff, dtype=np.uint8)


@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("mode", ["default", "lenient", "strict"])
def test_config_matches_oracle(name, mode):
    pol_fn, req_fn = CONFIGS[name]
    rt = rule_table_from_policies(policies_from_docs(pol_fn()))
    lt = lower_rule_table(rt)
    assert not lt.unsupported, lt.unsupported
    cr = req_fn(600)
    inputs = cr.to_inputs()
    kw = {"lenient_scope_search": mode == "lenient", "strict_evaluation": mode == "strict"}
    want = oracle_effects(rt, inputs, **kw)
    flags = (capi.F_LENIENT_SCOPE_SEARCH if mode == "lenient" else 0) | (capi.F_STRICT_EVALUATION if mode == "strict" else 0)
    fl = Flattener(lt)
    for batch in (fl.flatten(inputs), cr.to_batch(fl)):
        res = hostsim_api.check(lt, batch, NOW, flags)
        assert (res.status != capi.ST_UNSUPPORTED).all()
        assert res.effect.shape == want.shape
        mism = np.nonzero(res.effect != want)[0]
        assert mism.size == 0, "first mismatching tuple %s of %s/%s" % (mism[:5], name, mode)
    # the workload must exercise both outcomes
    assert 0 < (want == capi.EFFECT_ALLOW).sum() < want.size


@pytest.mark.parametrize("name,n", [("C3", 8000), ("C5", 8000)])
def test_kernel_source_matches_cpp_oracle_on_larger_batches(name, n):
    """The host-simulated kernels against oracle/ccheck.cpp (pinned on oracle/check.py in test_ccheck.py) on
    batches too large for the Python oracle: every output of every tuple ccheck covers (it flags requests
    that need general CEL programs)."""
    from oracle import ccheck
    pol_fn, req_fn = CONFIGS[name]
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol_fn())))
    batch = req_fn(n).to_batch(Flattener(lt))
    for flags in (capi.F_WANT_DERIVED_ROLES, capi.F_WANT_DERIVED_ROLES | capi.F_STRICT_EVALUATION,
                  capi.F_WANT_DERIVED_ROLES | capi.F_LENIENT_SCOPE_SEARCH):
        want = ccheck.check(lt, batch, NOW, flags, 4)
        got = hostsim_api.check(lt, batch, NOW, flags)
        ok = want.status != capi.ST_UNSUPPORTED
        assert ok.mean() > 0.7
        for f in ("effect", "policy", "scope"):
            assert np.array_equal(getattr(got, f)[ok], getattr(want, f)[ok]), (f, flags)
        okr = ok.reshape(-1, 4).all(axis=1)
        assert np.array_equal(got.edr[okr], want.edr[okr])
        assert np.array_equal((got.status == capi.ST_CEL_ERROR).reshape(-1, 4).any(axis=1)[okr],
                              (want.status == capi.ST_CEL_ERROR).reshape(-1, 4).any(axis=1)[okr])
