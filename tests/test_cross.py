"""CPU tier: a cross-product batch (N principals x M resources flattened once each) must decide exactly like the
N*M explicit CheckInputs - kernel source on the host simulator, Python flattener and C++ ingest."""
import numpy as np
import pytest

import hostsim_api
from cerbos_amd import capi
from cerbos_amd.cross import cross_product_batch, effect_cube, result_cubes
from cerbos_amd.flatten import Flattener
from cerbos_amd.ingest import WireFlattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.lower.celc import LoweringError
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from test_fuzz_parity import ACTIONS, NOW, _policies, _requests


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("native", [False, True])
def test_cross_product_equals_explicit_inputs(seed, native):
    rng = np.random.default_rng(30_000 + seed)
    try:
        lt = lower_rule_table(rule_table_from_policies(policies_from_docs(_policies(rng))))
    except LoweringError:
        pytest.skip("store refused by the lowering")
    sample = _requests(rng, 40)
    principals = [s["principal"] for s in sample[:17]]
    resources = [s["resource"] for s in sample[17:40]]
    aux = [s.get("auxData") for s in sample[:17]]
    actions = ACTIONS[:5]
    fl = WireFlattener(lt) if native else Flattener(lt)
    explicit = [dict({"principal": p, "resource": r, "actions": actions}, **({"auxData": x} if x else {}))
                for p, x in zip(principals, aux) for r in resources]
    for flags in (capi.F_WANT_DERIVED_ROLES, capi.F_WANT_DERIVED_ROLES | capi.F_LENIENT_SCOPE_SEARCH, capi.F_STRICT_EVALUATION):
        want = hostsim_api.check(lt, Flattener(lt).flatten(explicit), NOW, flags)
        cb = cross_product_batch(fl, lt.columns, principals, resources, actions, aux)
        got = hostsim_api.check(lt, cb, NOW, flags)
        cubes, edr = result_cubes(cb, got)
        for f in ("effect", "policy", "scope", "status"):
            assert np.array_equal(cubes[f].reshape(-1), getattr(want, f)), (f, flags)      # explicit inputs are principal-major
        assert np.array_equal(edr.reshape(-1), want.edr)
        assert np.array_equal(effect_cube(cb, got), cubes["effect"]) and cubes["effect"].shape == (17, 23, 5)
        # and the device order is the routing order the kernels like
        from cerbos_amd.flatten import RQ_KIND, RQ_R_SCOPE, RQ_R_VERSION, RQ_ROLE_CNT
        route = list(zip(*(cb.req_u32[f].tolist() for f in (RQ_KIND, RQ_R_VERSION, RQ_R_SCOPE))))
        assert route == sorted(route)                                   # resources ordered by route ...
        cnt = cb.req_u32[RQ_ROLE_CNT].reshape(23, 17)
        assert (np.diff(cnt.astype(np.int64), axis=1) >= 0).all()       # ... and within one resource, principals by role list
