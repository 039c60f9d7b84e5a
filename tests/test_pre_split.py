"""The walk's pre-pass in two kernels (CBH_PRE_SPLIT=1: cbh_walk2_collect_kernel appends every (request, site) the walk meets to the
site's list, cbh_walk2_interp_kernel evaluates the lists on full waves) on the simulator: the same decisions, statuses, policies, scopes
and derived roles as the fused pre-pass, on C5 (both request shapes), the engine goldens and generated stores.  Off by default: it is
to be measured first."""
import os

import numpy as np
import pytest

import hostsim_api
from cerbos_amd import capi, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.lower.celc import LoweringError
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json, store_rule_table
from test_fuzz_parity import _policies, _requests
from test_hostsim_golden import GLOBALS

NOW = 1_700_000_000_000_000_000
FIELDS = ("effect", "policy", "scope", "status", "edr")


def _both(lt, batch, flags=capi.F_WANT_DERIVED_ROLES):
    os.environ.pop("CBH_PRE_SPLIT", None)
    want = hostsim_api.check(lt, batch, NOW, flags, device_order=True)
    fused_kind = hostsim_api.last_kind()
    os.environ["CBH_PRE_SPLIT"] = "1"
    try:
        have = hostsim_api.check(lt, batch, NOW, flags, device_order=True)
        split = hostsim_api.last_pre_split()
    finally:
        os.environ.pop("CBH_PRE_SPLIT", None)
    for f in FIELDS:
        a, b = getattr(want, f), getattr(have, f)
        assert np.array_equal(a, b), (f, int(np.flatnonzero(a != b)[0]))
    return fused_kind, split


@pytest.mark.parametrize("req_fn", [workloads.c5_requests, workloads.c5w_requests], ids=["C5", "C5W"])
def test_c5_by_both_pre_passes(req_fn):
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
    batch = req_fn(3000).to_batch(Flattener(lt))
    for flags in (capi.F_WANT_DERIVED_ROLES, capi.F_WANT_DERIVED_ROLES | capi.F_LENIENT_SCOPE_SEARCH):
        kind, split = _both(lt, batch, flags)
        assert kind == 2 and split


def test_the_golden_store_by_both_pre_passes():
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    inputs = [i for c in load_json("engine_cases.json") for i in c["inputs"]]
    batch = Flattener(lt).flatten(inputs, "default", "")
    kind, split = _both(lt, batch)
    assert kind == 2
    # (the store's programs read runtime.effectiveDerivedRoles: such tables keep the fused pre-pass - their sites need the scope's roles first)
    assert split == (not lt.uses_runtime_edr) if hasattr(lt, "uses_runtime_edr") else True


@pytest.mark.parametrize("seed", range(10))
def test_generated_stores_by_both_pre_passes(seed):
    rng = np.random.default_rng(300 + seed)
    try:
        lt = lower_rule_table(rule_table_from_policies(policies_from_docs(_policies(rng))))
    except LoweringError:
        pytest.skip("store outside the device subset")
    batch = Flattener(lt).flatten(_requests(rng, 400), "default", "")
    _both(lt, batch)
    _both(lt, batch, capi.F_WANT_DERIVED_ROLES | capi.F_LENIENT_SCOPE_SEARCH)
