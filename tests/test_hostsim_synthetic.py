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
