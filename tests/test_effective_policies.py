"""AuditTrail.EffectivePolicies (engine.go:289-338, check.go:302-304) on the simulator: cbh_check_batch_trail's masks -> keys, against
the reference's own decision logs (the engine goldens: the call's union), against the oracle input by input (goldens, fuzz stores,
the synthetic configurations), and the decisions of that walk against the ordinary road's."""
import numpy as np
import pytest

import hostsim_api
from cerbos_amd import capi, workloads
from cerbos_amd.engine import Conf, effective_policy_keys
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json, store_rule_table
from oracle.check import EvalParams, RuleTableOracle
from test_fuzz_parity import _policies, _requests
from test_hostsim_golden import GLOBALS, HostSimEvaluator

NOW = 1_700_000_000_000_000_000
CASES = [c for c in load_json("engine_cases.json") if c["hasDecisionLogs"] and not c["wantError"]]


@pytest.fixture(scope="module")
def store():
    rt = store_rule_table()
    lt = lower_rule_table(rt, GLOBALS)
    return HostSimEvaluator(lt, Conf(globals_=GLOBALS)), RuleTableOracle(rt)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_the_reference_s_decision_logs(store, case):
    ev, oracle = store
    for lenient in ([False, True] if case["lenient"] is None else [case["lenient"]]):
        have = ev.effective_policies(case["inputs"], now_ns=NOW, lenient_scope_search=lenient)
        assert have == case["wantEffectivePolicies"], (case["name"], lenient)
        per_input = ev.effective_policies(case["inputs"], now_ns=NOW, lenient_scope_search=lenient, per_input=True)
        params = EvalParams(globals_=GLOBALS, now_ns=NOW, lenient_scope_search=lenient)
        assert per_input == [oracle.check(i, params)["effectivePolicies"] for i in case["inputs"]]


@pytest.mark.parametrize("strict", [False, True])
def test_all_golden_inputs_in_one_batch_by_input(store, strict):
    """Every input of every case as ONE device batch, grouped per input (what a coalescer in front of many calls does)."""
    ev, oracle = store
    inputs = [i for c in CASES for i in c["inputs"]]
    for lenient in (False, True):
        have = ev.effective_policies(inputs, now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict, per_input=True)
        params = EvalParams(globals_=GLOBALS, now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict)
        want = [oracle.check(i, params)["effectivePolicies"] for i in inputs]
        assert have == want, [k for k in range(len(inputs)) if have[k] != want[k]][:5]


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_stores_by_input(seed):
    """Generated stores (scopes, scope permissions, derived roles, role policies with parent roles, principal policies, globs)."""
    rng = np.random.default_rng(100 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    lt = lower_rule_table(rt)
    ev, oracle = HostSimEvaluator(lt, Conf()), RuleTableOracle(rt)
    inputs = _requests(rng, 300)
    for lenient in (False, True):
        have = ev.effective_policies(inputs, now_ns=NOW, lenient_scope_search=lenient, per_input=True)
        params = EvalParams(now_ns=NOW, lenient_scope_search=lenient)
        want = [oracle.check(i, params)["effectivePolicies"] for i in inputs]
        bad = [k for k in range(len(inputs)) if have[k] != want[k]]
        assert not bad, (seed, lenient, bad[:3], have[bad[0]], want[bad[0]], inputs[bad[0]])
    assert any(len(k) > 1 for k in have)


@pytest.mark.parametrize("general", [False, True], ids=["cbh_walk2_trail_kernel", "the general walk"])
def test_the_trail_s_walk_decides_like_the_ordinary_road(store, general, monkeypatch):
    """The golden store is cbh_walk2_kernel's: its trail form walks twice (w2_body EP) and answers - status included - what the
    ordinary walk answers; CBH_NO_WALK2=1: the general walk's trail form (what tables outside the walk's shapes take)."""
    ev, _ = store
    lt = ev.lt
    inputs = [i for c in CASES for i in c["inputs"]]
    batch = Flattener(lt).flatten(inputs, "default", "")
    if general:
        monkeypatch.setenv("CBH_NO_WALK2", "1")
    want = hostsim_api.check(lt, batch, NOW, capi.F_WANT_DERIVED_ROLES, device_order=True)
    have, masks = hostsim_api.check_trail(lt, batch, None, 1, NOW, capi.F_WANT_DERIVED_ROLES)
    assert hostsim_api.last_kind() == (0 if general else 2)
    for f in ("effect", "policy", "scope", "edr", "status"):
        assert np.array_equal(getattr(have, f), getattr(want, f)), f
    # one group: the union of what the cases that run without lenient scope search log
    want_keys = {k for c in CASES if not c["lenient"] for k in c["wantEffectivePolicies"]}
    assert want_keys <= set(effective_policy_keys(lt.policy_keys, masks[0]))


@pytest.mark.parametrize("env", [{}, {"CBH_NO_WALK2": "1"}, {"CBH_NO_WALK2_WIDE": "1"}], ids=["walk2 trail", "general trail", "walk2 trail, wide requests to the general walk"])
@pytest.mark.parametrize("name,n", [("c5", 600), ("c5w", 300), ("c5aw", 300)])
def test_walk_tables_by_input(name, n, env, monkeypatch):
    """C5 / C5W (scopes, derived roles, role policies, conditions with evaluation sites; C5W: five to eight roles, nine to sixteen
    actions - the walk's wider forms) input by input against the oracle, through the walk's trail kernels and the general walk's."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rt = rule_table_from_policies(policies_from_docs(workloads.c5_policies()))   # (C5W: C5's policies, wider requests)
    ev, oracle = HostSimEvaluator(lower_rule_table(rt), Conf()), RuleTableOracle(rt)
    if name == "c5aw":   # nine to sixteen actions: every third input asks for twelve of the actions the workload uses
        inputs = workloads.c5_requests(n_requests=n).to_inputs()
        pool = sorted({a for i in inputs for a in i["actions"]}) + ["archive", "export", "print:public", "comment"]   # (+ some no rule names)
        for k in range(0, n, 3):
            inputs[k] = dict(inputs[k], actions=[pool[(k + j) % len(pool)] for j in range(9 + k % 4)])
    else:
        inputs = getattr(workloads, name + "_requests")(n_requests=n).to_inputs()
    have = ev.effective_policies(inputs, now_ns=NOW, per_input=True)
    if name != "c5" and "CBH_NO_WALK2" not in env:
        assert hostsim_api.last_walk_wide() == (0 if "CBH_NO_WALK2_WIDE" in env else 1 if name == "c5w" else 2)
    assert hostsim_api.last_kind() == (0 if "CBH_NO_WALK2" in env else 2)
    params = EvalParams(now_ns=NOW)
    want = [oracle.check(i, params)["effectivePolicies"] for i in inputs]
    bad = [k for k in range(n) if have[k] != want[k]]
    assert not bad, (bad[:3], have[bad[0]], want[bad[0]], inputs[bad[0]])
    assert len({tuple(k) for k in have}) > 3


# ---- flat tables: the trail from the fast kernels (flat_body EP: scalar, staged and mask walks), not the general walk


@pytest.mark.parametrize("env", [{}, {"CBH_FLAT_MASKS": "0"}, {"CBH_FLAT_ANY": "1"}, {"CBH_FLAT_ANY": "1", "CBH_FLAT_MASKS": "0"}, {"CBH_FLAT_MASKS": "1"}],
                         ids=["as planned", "staged", "with the evaluator call", "staged with the call", "masks forced"])
@pytest.mark.parametrize("name,n", [("c2", 300), ("c3", 400), ("c4", 40), ("t", 90)])
def test_flat_tables_keep_their_trail_in_the_flat_kernels(name, n, env, monkeypatch):
    if name == "c4" and env != {}:
        pytest.skip("C4's fifty thousand rules once, as planned (the mask walk): T covers the other variants")
    if name == "t" and env == {"CBH_FLAT_ANY": "1", "CBH_FLAT_MASKS": "0"}:
        pytest.skip("the staged walk with the evaluator call: C2 and C3 cover it")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rt = rule_table_from_policies(policies_from_docs(getattr(workloads, name + "_policies")()))
    lt = lower_rule_table(rt)
    ev, oracle = HostSimEvaluator(lt, Conf()), RuleTableOracle(rt)
    inputs = getattr(workloads, name + "_requests")(n_requests=n).to_inputs()
    for lenient in (False, True):
        have = ev.effective_policies(inputs, now_ns=NOW, lenient_scope_search=lenient, per_input=True)
        assert hostsim_api.last_kind() == 1, "a flat kernel decides this"
        params = EvalParams(now_ns=NOW, lenient_scope_search=lenient)
        want = [oracle.check(i, params)["effectivePolicies"] for i in inputs]
        bad = [k for k in range(len(inputs)) if have[k] != want[k]]
        assert not bad, (name, lenient, bad[:3], have[bad[0]], want[bad[0]], inputs[bad[0]])
    # ... and decides as the ordinary flat kernel does
    batch = Flattener(lt).flatten(inputs, "default", "")
    want_res = hostsim_api.check(lt, batch, NOW, capi.F_WANT_DERIVED_ROLES, device_order=True)
    have_res, _ = hostsim_api.check_trail(lt, batch, None, 1, NOW, capi.F_WANT_DERIVED_ROLES)
    for f in ("effect", "policy", "scope", "status", "edr"):
        assert np.array_equal(getattr(have_res, f), getattr(want_res, f)), f
