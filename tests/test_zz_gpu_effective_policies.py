"""GPU tier (sorts last - written after the round's GPU minutes were spent): cbh_check_batch_trail through the C ABI -
AuditTrail.EffectivePolicies against the reference's decision logs (engine goldens), against the oracle input by input (goldens,
fuzz stores, C5 at size) and the trail walk's decisions against cbh_check_batch's."""
import numpy as np
import pytest

from cerbos_amd import capi, workloads
from cerbos_amd.engine import Conf, HipEvaluator, effective_policy_keys
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json, store_rule_table
from oracle.check import EvalParams, RuleTableOracle
from test_fuzz_parity import _policies, _requests
from test_hostsim_golden import GLOBALS

pytestmark = pytest.mark.gpu
NOW = 1_700_000_000_000_000_000
CASES = [c for c in load_json("engine_cases.json") if c["hasDecisionLogs"] and not c["wantError"]]


def test_the_reference_s_decision_logs_and_the_oracle_by_input():
    rt = store_rule_table()
    ev, oracle = HipEvaluator(lower_rule_table(rt, GLOBALS), Conf(globals_=GLOBALS)), RuleTableOracle(rt)
    try:
        assert ev.table.policy_keys() == ev.lt.policy_keys
        for case in CASES:
            for lenient in ([False, True] if case["lenient"] is None else [case["lenient"]]):
                assert ev.effective_policies(case["inputs"], now_ns=NOW, lenient_scope_search=lenient) == case["wantEffectivePolicies"], case["name"]
        inputs = [i for c in CASES for i in c["inputs"]]
        for lenient, strict in ((False, False), (True, False), (False, True)):
            have = ev.effective_policies(inputs, now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict, per_input=True)
            params = EvalParams(globals_=GLOBALS, now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict)
            assert have == [oracle.check(i, params)["effectivePolicies"] for i in inputs]
    finally:
        ev.close()


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_stores_by_input(seed):
    rng = np.random.default_rng(100 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    ev, oracle = HipEvaluator(lower_rule_table(rt), Conf()), RuleTableOracle(rt)
    try:
        inputs = _requests(rng, 600)
        have = ev.effective_policies(inputs, now_ns=NOW, per_input=True)
        params = EvalParams(now_ns=NOW)
        assert have == [oracle.check(i, params)["effectivePolicies"] for i in inputs]
    finally:
        ev.close()


def test_c5_at_size_groups_and_decisions(n=20_000):
    rt = rule_table_from_policies(policies_from_docs(workloads.c5_policies()))
    lt = lower_rule_table(rt)
    inputs = workloads.c5_requests(n_requests=n).to_inputs()
    batch = Flattener(lt).flatten(inputs, "default", "")
    pre = np.arange(batch.n_requests) if batch.req_perm is None else np.asarray(batch.req_perm)
    groups = (np.asarray(batch.vreq_input)[pre] % 97).astype(np.uint32)          # 97 "calls" mixed through the batch
    table = capi.Table(lt.blob)
    try:
        want = table.check(batch, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES, device_order=True)
        have, masks = table.check_trail(batch, groups, 97, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        for f in ("effect", "policy", "scope", "edr", "status"):   # (cbh_walk2_trail_kernel: the second walk's results are the first's)
            assert np.array_equal(getattr(have, f), getattr(want, f)), f
        oracle, params = RuleTableOracle(rt), EvalParams(now_ns=NOW)
        want_keys = [set() for _ in range(97)]
        for k in range(0, n, 7):                                                  # a seventh of the inputs through the oracle: a lower bound per group
            want_keys[k % 97].update(oracle.check(inputs[k], params)["effectivePolicies"])
        for g in range(97):
            assert want_keys[g] <= set(effective_policy_keys(lt.policy_keys, masks[g]))
        every = set().union(*(set(effective_policy_keys(lt.policy_keys, m)) for m in masks))
        assert every <= set(lt.policy_keys) and len(every) > 1
    finally:
        table.close()


@pytest.mark.parametrize("name,n", [("c2", 20_000), ("c3", 20_000), ("t", 6_000)])
def test_flat_tables_keep_their_trail_in_the_flat_kernels(name, n):
    """flat_body EP (cbh_check_flat_trail_kernel*): the trail from the fast kernels - per input against the oracle, decisions as cbh_check_batch's"""
    rt = rule_table_from_policies(policies_from_docs(getattr(workloads, name + "_policies")()))
    lt = lower_rule_table(rt)
    inputs = getattr(workloads, name + "_requests")(n_requests=n).to_inputs()
    ev, oracle = HipEvaluator(lt, Conf()), RuleTableOracle(rt)
    try:
        have = ev.effective_policies(inputs, now_ns=NOW, per_input=True)
        params = EvalParams(now_ns=NOW)
        step = max(1, n // 3000)
        for k in range(0, n, step):
            assert have[k] == oracle.check(inputs[k], params)["effectivePolicies"], (name, k)
        batch = Flattener(lt).flatten(inputs, "default", "")
        want = ev.table.check(batch, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES, device_order=True)
        got, masks = ev.table.check_trail(batch, None, 1, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        for f in ("effect", "policy", "scope", "edr"):
            assert np.array_equal(getattr(got, f), getattr(want, f)), f
        assert set(effective_policy_keys(lt.policy_keys, masks[0])) == set().union(*map(set, have))
    finally:
        ev.close()


@pytest.mark.parametrize("name,n", [("c5", 3_000), ("c5w", 1_500)])
def test_walk_tables_by_input(name, n):
    """cbh_walk2_trail_kernel and its wider forms (w2_body EP: the walk twice, the second one marks) input by input against the oracle."""
    rt = rule_table_from_policies(policies_from_docs(workloads.c5_policies()))
    ev, oracle = HipEvaluator(lower_rule_table(rt), Conf()), RuleTableOracle(rt)
    try:
        inputs = getattr(workloads, name + "_requests")(n_requests=n).to_inputs()
        if name == "c5w":   # ... and some with nine to twelve actions
            pool = sorted({a for i in inputs for a in i["actions"]}) + ["archive", "export", "print:public", "comment"]
            for k in range(0, n, 5):
                inputs[k] = dict(inputs[k], actions=[pool[(k + j) % len(pool)] for j in range(9 + k % 4)],
                                 principal=dict(inputs[k]["principal"], roles=inputs[k]["principal"]["roles"][:3]))   # (at most three roles: the nine-to-sixteen-action form)
        have = ev.effective_policies(inputs, now_ns=NOW, per_input=True)
        params = EvalParams(now_ns=NOW)
        want = [oracle.check(i, params)["effectivePolicies"] for i in inputs]
        bad = [k for k in range(n) if have[k] != want[k]]
        assert not bad, (bad[:3], have[bad[0]], want[bad[0]])
    finally:
        ev.close()


def test_a_resident_batch_keeps_its_trail_across_launches(n=5_000):
    """cbh_batch_set_trail / cbh_trail_download: the masks of a resident batch are what cbh_check_batch_trail returns for it, launch
    after launch (OR-ed: the same), cleared by the next cbh_batch_set_trail; the trail kernels are the ones the plan names."""
    rt = rule_table_from_policies(policies_from_docs(workloads.c5_policies()))
    lt = lower_rule_table(rt)
    inputs = workloads.c5_requests(n_requests=n).to_inputs()
    batch = Flattener(lt).flatten(inputs, "default", "")
    groups = (np.arange(batch.n_requests) % 11).astype(np.uint32)
    table = capi.Table(lt.blob)
    try:
        _, want = table.check_trail(batch, groups, 11, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
        db = table.upload(batch)
        flags = capi.F_WANT_DERIVED_ROLES | capi.F_WANT_EFFECTIVE_POLICIES
        assert "trail" in table.plan(db, flags)
        table.set_trail(db, groups, 11)
        for _ in range(3):
            table.launch(db, now_ns=NOW, flags=flags)
        assert np.array_equal(table.trail(db), want)
        table.set_trail(db, None, 1)
        assert not table.trail(db).any()
        table.launch(db, now_ns=NOW, flags=flags)
        assert np.array_equal(table.trail(db)[0], np.bitwise_or.reduce(want, axis=0))
        plain = table.download(db)
        ordinary = table.check(batch, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)   # (both in input order)
        assert np.array_equal(plain.effect, ordinary.effect) and np.array_equal(plain.status, ordinary.status)
        db.close()
    finally:
        table.close()
