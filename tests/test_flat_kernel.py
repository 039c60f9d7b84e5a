"""cbh_check_flat_kernel (cbh_check_flat.h): the scopes -> records -> roles-side-by-side walk for flat tables, fuzzed
against oracle/check.py.

Stores here are built to BE flat (resource policies only, literal or `*` roles and actions, leaf conditions) and to
stress what the inside-out loop order must keep exact: several roles per principal where a later role allows at a
shallower scope than an earlier one, DENY rules between ALLOW rules, REQUIRE_PARENTAL_CONSENT / OVERRIDE_PARENT
chains, conditions that raise CEL errors (missing attributes) on rules only a LATER role matches - the reference
never evaluates a role after the one that allowed (check.go:433-436), so those errors must not be reported.
Compared per action: effect, policy key, scope; per request: whether evaluation errors were recorded.
CPU tier: the kernel source on the host simulator.  GPU tier: the kernel.
"""
import numpy as np
import pytest

from cerbos_amd import capi
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000
SCOPES = ["", "acme", "acme.hr", "acme.hr.uk"]
KINDS = ["doc", "report"]
ROLES = ["user", "manager", "admin", "guest", "auditor", "intern"]
ACTIONS = ["view", "edit", "delete", "approve", "share", "export"]
# conditions that leave the classified leaves' inline code: container equality (class 3 on lists), a list where a
# number is expected - these batches carry list tags and run the kernel variant with the evaluator call
CONDS_ANY = ["R.attr.tags == P.attr.tags", "R.attr.tags != P.attr.tags", "R.attr.amount >= 50"]
CONDS = ["R.attr.public == true", "R.attr.owner == P.id", "R.attr.amount > 100", "P.attr.department == R.attr.department",
         'R.attr.status in ["OPEN", "PENDING"]', "R.attr.missing == 1", "P.attr.level >= 3", "R.attr.owner != P.id"]


def _cond(rng, pool=None):
    pool = pool or CONDS
    r = rng.random()
    if r < 0.35:
        return None
    if r < 0.8:
        return {"match": {"expr": str(rng.choice(pool))}}
    kind = str(rng.choice(["all", "any", "none"]))
    of = [{"expr": str(e)} for e in rng.choice(pool, size=int(rng.integers(2, 4)), replace=False)]
    if rng.random() < 0.3:   # a nested level
        inner = str(rng.choice(["all", "any", "none"]))
        of.append({inner: {"of": [{"expr": str(e)} for e in rng.choice(pool, size=2, replace=False)]}})
    return {"match": {kind: {"of": of}}}


DRS = ["owner", "peer", "senior", "anyone"]


def _store(rng, pool=None, many_rules=False, deep=False):
    docs = []
    scopes = SCOPES + (["acme.hr.uk.london", "acme.hr.uk.london.e1"] if deep else [])
    with_dr = rng.random() < 0.6
    if with_dr:   # derived roles: definitions with and without conditions, `*` parents, conditions that can raise errors
        docs.append({"apiVersion": API, "derivedRoles": {"name": "flat_roles", "definitions": [
            {"name": "owner", "parentRoles": [str(r) for r in rng.choice(ROLES, size=2, replace=False)],
             "condition": {"match": {"expr": "R.attr.owner == P.id"}}},
            {"name": "peer", "parentRoles": ["*"], "condition": {"match": {"expr": "P.attr.department == R.attr.department"}}},
            {"name": "senior", "parentRoles": [str(r) for r in rng.choice(ROLES, size=3, replace=False)],
             "condition": {"match": {"expr": str(rng.choice(["P.attr.level >= 3", "R.attr.missing == 1"]))}}},
            {"name": "anyone", "parentRoles": [str(rng.choice(ROLES))]}]}})
    for kind in KINDS:
        for si, scope in enumerate(scopes):
            if scope and rng.random() < 0.2:
                continue
            rules = []
            for _ in range(int(rng.integers(14, 80)) if many_rules else int(rng.integers(1, 7))):
                rule = {"actions": [str(a) for a in rng.choice(ACTIONS + ["*"], size=int(rng.integers(1, 6)), replace=False)],
                        "effect": "EFFECT_ALLOW" if rng.random() < 0.7 else "EFFECT_DENY"}
                if with_dr and rng.random() < 0.4:
                    rule["derivedRoles"] = [str(d) for d in rng.choice(DRS, size=int(rng.integers(1, 3)), replace=False)]
                    if rng.random() < 0.3:
                        rule["roles"] = [str(rng.choice(ROLES))]
                else:
                    rule["roles"] = [str(r) for r in rng.choice(ROLES + ["*"], size=int(rng.integers(1, 5)), replace=False)]
                c = _cond(rng, pool)
                if c:
                    rule["condition"] = c
                rules.append(rule)
            pol = {"resource": kind, "version": "default", "rules": rules}
            if with_dr:
                pol["importDerivedRoles"] = ["flat_roles"]
            if scope:
                pol["scope"] = scope
                if rng.random() < 0.5:
                    pol["scopePermissions"] = "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS"
            docs.append({"apiVersion": API, "resourcePolicy": pol})
    return docs


def _requests(rng, n, with_lists=False, deep=False):
    out = []
    req_scopes = SCOPES + ["acme.hr.uk.london", "zzz"] + (["acme.hr.uk.london.e1", "acme.hr.uk.london.e1.desk4"] if deep else [])
    for i in range(n):
        roles = [str(r) for r in rng.choice(ROLES + ["stranger"], size=int(rng.integers(0, 5)), replace=False)]
        acts = [str(a) for a in rng.choice(ACTIONS + ["nothing"], size=int(rng.integers(0, 5)), replace=False)]
        pid = "p%d" % rng.integers(0, 4)
        rattr = {"public": bool(rng.random() < 0.4), "owner": "p%d" % rng.integers(0, 4), "amount": float(rng.integers(0, 200)),
                 "department": str(rng.choice(["eng", "ops"])), "status": str(rng.choice(["OPEN", "CLOSED"]))}
        pattr = {"department": str(rng.choice(["eng", "ops"])), "level": float(rng.integers(1, 6))}
        if with_lists:
            tagsets = [["a"], ["a", "b"], ["b", "a"], [], "a", 3.0]
            rattr["tags"] = tagsets[int(rng.integers(0, len(tagsets)))]
            pattr["tags"] = tagsets[int(rng.integers(0, len(tagsets)))]
            if rng.random() < 0.2:
                rattr["amount"] = [1.0, 2.0]   # a list where the policies compare a number
        for d in (rattr, pattr):
            for k in list(d):
                if rng.random() < 0.08:
                    del d[k]
        out.append({"requestId": "q%d" % i, "actions": acts, "principal": {"id": pid, "roles": roles, "attr": pattr},
                    "resource": {"kind": str(rng.choice(KINDS + ["other"])), "id": "r%d" % i, "attr": rattr,
                                 "scope": str(rng.choice(req_scopes))}})
    return out


def _run_seed(seed, make_evaluator, close, with_lists=False, many_rules=False, deep=False):
    rng = np.random.default_rng(40_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_store(rng, CONDS + CONDS_ANY if with_lists else None, many_rules, deep)))
    lt = lower_rule_table(rt)
    assert lt.stats["flat"] and lt.stats["flat_closed"], "the generator must produce flat tables whose conditions are all inline"
    inputs = _requests(rng, 200, with_lists, deep)
    batch = Flattener(lt).flatten(inputs)
    plain = not np.isin(batch.col_tag, (2, 3, 6, 7)).any()   # cbh_engine.hip validate_batch: selects the kernel variant
    assert plain != with_lists
    assert int(batch.req_u32[9].max()) <= 4 and int(batch.req_u32[7].max()) <= 4   # the flat kernel's batch shape
    ev = make_evaluator(lt)
    orc = RuleTableOracle(rt)
    n_err = 0
    try:
        for lenient in (False, True):
            flags = capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0)
            res = ev.table.check(batch, now_ns=NOW, flags=flags)
            outs, bad = ev.assemble(inputs, batch, res, "default", allow_unsupported=True)
            assert not bad
            t = 0
            for inp, have in zip(inputs, outs):
                want = orc.check(inp, EvalParams(now_ns=NOW, lenient_scope_search=lenient))
                assert norm_actions(have) == norm_actions(want), (seed, lenient, inp)
                assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or []), (seed, lenient, inp)
                na = len(inp["actions"])
                got_err = bool((res.status[t:t + na] == capi.ST_CEL_ERROR).any())
                assert got_err == bool(want.get("evaluationErrors")), (seed, lenient, inp, want.get("evaluationErrors"))
                n_err += got_err
                t += na
    finally:
        if close:
            ev.close()
    return n_err


@pytest.mark.parametrize("seed", range(40))
def test_flat_kernel_source_vs_oracle(seed):
    from test_hostsim_golden import HostSimEvaluator
    _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.parametrize("seed", range(100, 116))
def test_flat_kernel_with_evaluator_call_vs_oracle(seed):
    """Batches with list-valued attributes: container equality and a list met where a number is compared go through
    the shared evaluator (cbh_check_flat_kernel_any)."""
    from test_hostsim_golden import HostSimEvaluator
    _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False, with_lists=True)


@pytest.mark.parametrize("seed", range(200, 212))
def test_flat_kernel_large_buckets_vs_oracle(seed):
    """Policies of 14-80 rules: tables with a bucket above CBH_FLAT_STAGE_MIN walk their records staged, 64 at a time."""
    from test_hostsim_golden import HostSimEvaluator
    _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False, many_rules=True)


@pytest.mark.parametrize("seed", range(300, 310))
def test_flat_kernel_deep_chains_vs_oracle(seed):
    """Scope chains of up to six entries: derived-role definitions are evaluated in the second climb (tables whose
    chains have at most four entries evaluate them in the walk itself)."""
    from test_hostsim_golden import HostSimEvaluator
    _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False, deep=True)


def test_plain_batches_through_the_variant_with_the_call(monkeypatch):
    """CBH_FLAT_ANY=1 sends plain batches through the variant with the call too: both variants decide them alike."""
    from test_hostsim_golden import HostSimEvaluator
    monkeypatch.setenv("CBH_FLAT_ANY", "1")
    for seed in range(6):
        _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False)


def test_staged_record_walk_on_any_table(monkeypatch):
    """CBH_FORCE_STAGED=1: the staged record walk (cbh_check_flat_kernel_staged / _any_staged: a bucket's records fetched
    64 at a time into the lanes' registers, candidates picked by a ballot over the class masks) on stores of every size,
    both variants, deep chains included - alike the scalar walk, which the other tests of this file pin."""
    from test_hostsim_golden import HostSimEvaluator
    monkeypatch.setenv("CBH_FORCE_STAGED", "1")
    for seed in range(8):
        _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False)
    for seed in range(100, 104):
        _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False, with_lists=True)
    for seed in range(200, 204):
        _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False, many_rules=True)
    _run_seed(300, lambda lt: HostSimEvaluator(lt, Conf()), False, deep=True)


def test_mask_walk_on_any_table(monkeypatch):
    """CBH_FLAT_MASKS=1: the mask walk (cbh_check_flat_kernel_masks / _any_masks: a bucket's segments decide by bitmaps per
    action class, role class and distinct condition - no record is visited) on stores of every size, both variants, deep
    chains and derived-role conditions included; tables with long buckets take it by themselves (the large-bucket tests)."""
    import hostsim_api
    from test_hostsim_golden import HostSimEvaluator
    monkeypatch.setenv("CBH_FLAT_MASKS", "1")
    for seed in range(12):
        _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False)
    for seed in range(100, 106):
        _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False, with_lists=True)
    for seed in range(300, 303):
        _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False, deep=True)
    assert hostsim_api.lib().hostsim_last_masks() == 1


def test_long_buckets_take_the_mask_walk_and_the_staged_walk_agrees(monkeypatch):
    import hostsim_api
    from test_hostsim_golden import HostSimEvaluator
    _run_seed(200, lambda lt: HostSimEvaluator(lt, Conf()), False, many_rules=True)
    assert hostsim_api.lib().hostsim_last_masks() == 1
    monkeypatch.setenv("CBH_FLAT_MASKS", "0")
    for seed in range(200, 204):
        _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False, many_rules=True)
    assert hostsim_api.lib().hostsim_last_masks() == 0


def test_segments_of_a_table_that_is_not_pooled():
    """More than 64 distinct fused leaves: every segment carries its own leaves and numbers (CBH_MSEG_POOLED off); a
    bucket of 150 rules is three segments."""
    import hostsim_api
    rng = np.random.default_rng(77)
    rules = []
    for i in range(150):
        rules.append({"actions": [str(a) for a in rng.choice(ACTIONS, size=2, replace=False)], "roles": [str(r) for r in rng.choice(ROLES, size=2, replace=False)],
                      "effect": "EFFECT_ALLOW" if rng.random() < 0.8 else "EFFECT_DENY",
                      "condition": {"match": {"expr": "R.attr.amount > %d" % i} if i % 3 else {"any": {"of": [{"expr": "R.attr.amount > %d" % (i + 200)}, {"expr": "R.attr.owner == P.id"}]}}}})
    docs = [{"apiVersion": API, "resourcePolicy": {"resource": "doc", "version": "default", "rules": rules}}]
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    assert lt.stats["flat"] and not lt.seg_stats["pooled"] and lt.seg_stats["segments"] == 3, lt.seg_stats
    inputs = [i for i in _requests(rng, 300, False, False) if i["resource"]["kind"] in ("doc", "other")]
    for i in inputs:
        i["resource"]["scope"] = ""
    batch = Flattener(lt).flatten(inputs)
    res = hostsim_api.check(lt, batch, NOW, capi.F_WANT_DERIVED_ROLES)
    assert hostsim_api.lib().hostsim_last_masks() == 1
    orc = RuleTableOracle(rt)
    t = 0
    for inp in inputs:
        want = orc.check(inp, EvalParams(now_ns=NOW))
        for a in inp["actions"]:
            assert (res.effect[t] == capi.EFFECT_ALLOW) == (want["actions"][a]["effect"] == "EFFECT_ALLOW"), (inp, a)
            t += 1
        na = len(inp["actions"])
        assert bool((res.status[t - na:t] == capi.ST_CEL_ERROR).any()) == bool(want.get("evaluationErrors"))


def test_error_cases_do_occur():
    from test_hostsim_golden import HostSimEvaluator
    assert sum(_run_seed(s, lambda lt: HostSimEvaluator(lt, Conf()), False) for s in range(4)) > 20


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_flat_kernel_on_gpu(seed):
    _run_seed(seed, lambda lt: HipEvaluator(lt, Conf()), True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(300, 310))
def test_flat_kernel_deep_chains_on_gpu(seed):
    _run_seed(seed, lambda lt: HipEvaluator(lt, Conf()), True, deep=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(200, 212))
def test_flat_kernel_large_buckets_on_gpu(seed):
    _run_seed(seed, lambda lt: HipEvaluator(lt, Conf()), True, many_rules=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 116))
def test_flat_kernel_with_evaluator_call_on_gpu(seed):
    _run_seed(seed, lambda lt: HipEvaluator(lt, Conf()), True, with_lists=True)


@pytest.mark.parametrize("seed", range(400, 404))
def test_flat_kernel_large_buckets_with_evaluator_call_vs_oracle(seed):
    """Long buckets AND list-valued attributes: cbh_check_flat_kernel_any_staged."""
    from test_hostsim_golden import HostSimEvaluator
    _run_seed(seed, lambda lt: HostSimEvaluator(lt, Conf()), False, with_lists=True, many_rules=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(400, 406))
def test_flat_kernel_large_buckets_with_evaluator_call_on_gpu(seed):
    _run_seed(seed, lambda lt: HipEvaluator(lt, Conf()), True, with_lists=True, many_rules=True)


@pytest.mark.parametrize("env", [{"CBH_NO_FLAT": "1"}, {"CBH_NO_FLAT": "1", "CBH_NO_WALK2": "1"}], ids=["cbh_walk2_kernel", "the general walk's leaf kernels"])
@pytest.mark.parametrize("name", ["c2", "c3"])
def test_flat_workloads_through_the_other_kernel_families(name, env, monkeypatch):
    """C2 / C3 are the flat kernels' - their classified leaves (a column against a constant, a column in a list of string constants,
    two columns) are also leaves of the walk and of the general walk's leaf kernels, which no other test sends these shapes through
    (the round's coverage table): the same answers, tuple by tuple, as the flat kernel and the oracle."""
    import hostsim_api
    from cerbos_amd import capi, workloads
    from cerbos_amd.flatten import Flattener
    from cerbos_amd.lower.blob import lower_rule_table
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies
    from oracle.check import EvalParams, RuleTableOracle
    rt = rule_table_from_policies(policies_from_docs(getattr(workloads, name + "_policies")()))
    lt = lower_rule_table(rt)
    cr = getattr(workloads, name + "_requests")(n_requests=400)
    batch = cr.to_batch(Flattener(lt))
    now = 1_700_000_000_000_000_000
    want = hostsim_api.check(lt, batch, now, capi.F_WANT_DERIVED_ROLES)
    assert hostsim_api.last_kind() == 1
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    have = hostsim_api.check(lt, batch, now, capi.F_WANT_DERIVED_ROLES)
    assert hostsim_api.last_kind() == (2 if len(env) == 1 else 0)
    for f in ("effect", "policy", "scope", "edr"):
        assert np.array_equal(getattr(have, f), getattr(want, f)), f
    assert ((have.status == capi.ST_UNSUPPORTED) == (want.status == capi.ST_UNSUPPORTED)).all()
    orc = RuleTableOracle(rt)
    effects = []
    for inp in cr.to_inputs()[:150]:
        out = orc.check(inp, EvalParams(now_ns=now))
        effects.extend(1 if out["actions"][a]["effect"] == "EFFECT_ALLOW" else 2 for a in inp["actions"])
    assert np.array_equal(have.effect[:len(effects)], np.array(effects, dtype=np.uint8))


def test_condition_trees_over_pooled_leaves_in_the_mask_walk():
    """A long bucket whose conditions are all / any / none trees over a handful of fused leaves (the table's leaves are POOLED: at most
    64 distinct ones, evaluated once per wave) - the mask walk folds a tree's level from its leaves' pooled codes; no other test builds
    that shape (the round's coverage table)."""
    import hostsim_api
    rng = np.random.default_rng(78)
    leaves = ["R.attr.amount > 500", "R.attr.owner == P.id", "R.attr.public == true", 'R.attr.status == "OPEN"', "R.attr.amount > 1500",
              'P.attr.department == R.attr.department']
    rules = []
    for i in range(90):
        a, b, c = (leaves[int(x)] for x in rng.choice(len(leaves), size=3, replace=False))
        kind = ("all", "any", "none")[i % 3]
        cond = {kind: {"of": [{"expr": a}, {"expr": b}]}} if i % 2 else {kind: {"of": [{"expr": a}, {"any": {"of": [{"expr": b}, {"expr": c}]}}]}}
        rules.append({"actions": [str(x) for x in rng.choice(ACTIONS, size=2, replace=False)], "roles": [str(r) for r in rng.choice(ROLES, size=2, replace=False)],
                      "effect": "EFFECT_ALLOW" if rng.random() < 0.8 else "EFFECT_DENY", "condition": {"match": cond}})
    docs = [{"apiVersion": API, "resourcePolicy": {"resource": "doc", "version": "default", "rules": rules}}]
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    assert lt.stats["flat"] and lt.seg_stats["pooled"], lt.seg_stats
    inputs = [i for i in _requests(rng, 300, False, False) if i["resource"]["kind"] in ("doc", "other")]
    for i in inputs:
        i["resource"]["scope"] = ""
    batch = Flattener(lt).flatten(inputs)
    res = hostsim_api.check(lt, batch, NOW, capi.F_WANT_DERIVED_ROLES)
    assert hostsim_api.lib().hostsim_last_masks() == 1
    orc = RuleTableOracle(rt)
    t = 0
    for inp in inputs:
        want = orc.check(inp, EvalParams(now_ns=NOW))
        for a in inp["actions"]:
            assert (res.effect[t] == capi.EFFECT_ALLOW) == (want["actions"][a]["effect"] == "EFFECT_ALLOW"), (inp, a)
            t += 1
        na = len(inp["actions"])
        assert bool((res.status[t - na:t] == capi.ST_CEL_ERROR).any()) == bool(want.get("evaluationErrors"))
