"""cbh_walk2_kernel (cbh_check_walk2.h): the scopes -> records -> roles-side-by-side walk for tables with principal
policies, role policies, parent roles, glob patterns and generic conditions - with its pre-pass for the evaluation
sites - against the general walk (cbh_check_wave.h, the same batch with CBH_NO_WALK2=1) tuple by tuple, and against
oracle/check.py per action (effect, policy, scope), per request (derived roles, whether evaluation errors were
recorded).  CPU tier: the kernel sources on the host simulator.  GPU tier: the kernels through the C ABI (oracle only:
the library reads its environment once)."""
import os

import numpy as np
import pytest

import hostsim_api
from cerbos_amd import capi, workloads
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.lower.celc import LoweringError
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions, store_rule_table
from oracle.check import EvalParams, RuleTableOracle
from test_fuzz_parity import _policies, _requests
from test_hostsim_golden import GLOBALS, HostSimEvaluator

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000


def _expr(e):
    return {"match": {"expr": e}}


class _NoWalk2:
    def __enter__(self):
        os.environ["CBH_NO_WALK2"] = "1"

    def __exit__(self, *exc):
        os.environ.pop("CBH_NO_WALK2", None)


def _both_kernels(lt, batch, flags):
    """The batch through cbh_walk2_kernel and through the general walk (device order): every output word must agree,
    the error status per request."""
    new = hostsim_api.check(lt, batch, NOW, flags, device_order=True)
    assert hostsim_api.last_kind() == 2, "the table / batch must run on cbh_walk2_kernel (refused: %s)" % lt.stats["walk2_refused"]
    _both_kernels.walk_wide = hostsim_api.last_walk_wide()   # did cbh_walk2_wide_kernel take part?
    with _NoWalk2():
        old = hostsim_api.check(lt, batch, NOW, flags, device_order=True)
        assert hostsim_api.last_kind() != 2
    for f in ("effect", "policy", "scope", "edr"):
        a, b = getattr(new, f), getattr(old, f)
        assert np.array_equal(a, b), (f, np.nonzero(a != b)[0][:8])
    assert np.array_equal(new.status == capi.ST_UNSUPPORTED, old.status == capi.ST_UNSUPPORTED)
    n = batch.n_requests
    off, cnt = batch.req_u32[8].astype(np.int64), batch.req_u32[9].astype(np.int64)
    for r in range(n):
        sl = slice(off[r], off[r] + cnt[r])
        assert (new.status[sl] == capi.ST_CEL_ERROR).any() == (old.status[sl] == capi.ST_CEL_ERROR).any(), r
    return new


def _against_oracle(rt, lt, make_evaluator, inputs, lenient=False, globals_=None):
    ev = make_evaluator(lt)
    try:
        batch = ev.flattener.flatten(inputs, "default", "")
        flags = capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0)
        res = ev.table.check(batch, now_ns=NOW, flags=flags)
        outs, bad = ev.assemble(inputs, batch, res, "default", allow_unsupported=True)
    finally:
        if not isinstance(ev, HostSimEvaluator):
            ev.close()
    orc = RuleTableOracle(rt)
    t = n_err = 0
    for i, (inp, have) in enumerate(zip(inputs, outs)):
        na = len(inp["actions"])
        if i not in bad:
            want = orc.check(inp, EvalParams(now_ns=NOW, lenient_scope_search=lenient, globals_=globals_))
            assert norm_actions(have) == norm_actions(want), (lenient, inp, have["actions"], want["actions"])
            assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or []), (lenient, inp)
            got_err = bool((res.status[t:t + na] == capi.ST_CEL_ERROR).any())
            assert got_err == bool(want.get("evaluationErrors")), (lenient, inp, want.get("evaluationErrors"))
            n_err += got_err
        t += na
    return len(inputs) - len(bad), n_err


def _hostsim(lt):
    return HostSimEvaluator(lt, Conf())


# ---------------------------------------------------------------------------------------------------------- lowering
def test_sites_and_slots():
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
    assert lt.stats["walk2"] and not lt.stats["flat"]
    generic, total = lt.stats["gslots"]
    assert 0 < generic <= total <= 256
    lt2 = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c1_policies())))
    assert not lt2.stats["walk2"] and lt2.stats["walk2_refused"]      # 120 literal actions: more than the class masks tell apart
    lt3 = lower_rule_table(store_rule_table(), GLOBALS)
    assert lt3.stats["walk2"], lt3.stats["walk2_refused"]             # the reference's own test store


# ---------------------------------------------------------------------------------------------------------- fuzz
@pytest.mark.parametrize("seed", range(40))
def test_fuzz_stores_walk2_vs_general_walk_vs_oracle(seed):
    rng = np.random.default_rng(77_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering (history-dependent reference behaviour)")
    if not lt.stats["walk2"]:
        pytest.skip("table stays on the general walk: %s" % lt.stats["walk2_refused"])
    inputs = _requests(rng, 220)
    batch = Flattener(lt).flatten(inputs)
    for lenient in (False, True):
        _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0))
        compared, _ = _against_oracle(rt, lt, _hostsim, inputs, lenient)
        assert compared > 120


# ---------------------------------------------------------------------------------------------------------- the wider shape
def _more_roles(rng, inputs, pool, share=0.5):
    """Give a share of the requests five to ten roles (the request's own first, then names of the pool and names no
    policy knows, shuffled): five to eight take cbh_walk2_wide_kernel, more than eight stay on the general walk."""
    out = []
    for inp in inputs:
        inp = dict(inp, principal=dict(inp["principal"]))
        if rng.random() < share:
            have = list(inp["principal"]["roles"])
            extra = [r for r in pool + ["nobody%d" % k for k in range(6)] if r not in have]
            rng.shuffle(extra)
            n = int(rng.choice([5, 6, 7, 8, 9, 10], p=[0.25, 0.2, 0.2, 0.2, 0.1, 0.05]))
            roles = have + extra[:max(0, n - len(have))]
            rng.shuffle(roles)
            inp["principal"]["roles"] = [str(r) for r in roles]
        out.append(inp)
    return out


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_stores_requests_with_five_to_eight_roles(seed):
    """The fuzz stores with requests of up to ten roles: cbh_walk2_wide_kernel (64-bit walk vectors) against the general
    walk tuple by tuple and against the oracle."""
    rng = np.random.default_rng(79_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering (history-dependent reference behaviour)")
    if not lt.stats["walk2"]:
        pytest.skip("table stays on the general walk: %s" % lt.stats["walk2_refused"])
    from test_fuzz_parity import ROLES
    inputs = _more_roles(rng, _requests(rng, 200), ROLES + ["other"])
    batch = Flattener(lt).flatten(inputs)
    assert 4 < int(batch.req_u32[7].max())
    for lenient in (False, True):
        _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0))
        assert _both_kernels.walk_wide & 1
        compared, _ = _against_oracle(rt, lt, _hostsim, inputs, lenient)
        assert compared > 100


def _more_actions(rng, inputs, pool, share=0.4):
    """Give a share of the requests nine to eighteen actions (their own first, then names of the pool and names no rule
    knows): nine to sixteen with at most four roles take cbh_walk2_awide_kernel, the rest of them the general walk."""
    out = []
    for inp in inputs:
        if rng.random() < share:
            have = list(inp["actions"])[:8]
            extra = [a for a in pool + ["nothing:%d" % k for k in range(12)] if a not in have]
            rng.shuffle(extra)
            n = int(rng.choice([9, 10, 12, 15, 16, 17, 18], p=[0.2, 0.15, 0.2, 0.15, 0.2, 0.05, 0.05]))
            acts = have + extra[:n - len(have)]
            rng.shuffle(acts)
            inp = dict(inp, actions=[str(a) for a in acts])
        out.append(inp)
    return out


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_stores_requests_with_nine_to_sixteen_actions(seed):
    """cbh_walk2_awide_kernel (16 actions x 4 roles) - and, in the same batches, the shape with more roles and the general walk
    for what neither holds - against the general walk tuple by tuple and against the oracle."""
    from test_fuzz_parity import ACTIONS, ROLES
    rng = np.random.default_rng(81_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering (history-dependent reference behaviour)")
    if not lt.stats["walk2"]:
        pytest.skip("table stays on the general walk: %s" % lt.stats["walk2_refused"])
    inputs = _more_actions(rng, _more_roles(rng, _requests(rng, 220), ROLES + ["other"], share=0.25), list(ACTIONS))
    batch = Flattener(lt).flatten(inputs)
    for lenient in (False, True):
        _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0))
        assert _both_kernels.walk_wide == 3
        compared, _ = _against_oracle(rt, lt, _hostsim, inputs, lenient)
        assert compared > 100


@pytest.mark.parametrize("seed", range(16))
def test_role_policy_chains_with_nine_to_sixteen_actions(seed):
    rng = np.random.default_rng(91_000 + seed)
    docs, acts = _role_policy_store(rng)
    rt = rule_table_from_policies(policies_from_docs(docs))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    inputs = _more_actions(rng, _role_policy_requests(rng, acts, 220), acts + ["view:x", "edit:y", "other"], share=0.5)
    batch = Flattener(lt).flatten(inputs)
    for lenient in (False, True):
        _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0))
        assert _both_kernels.walk_wide & 2
        _against_oracle(rt, lt, _hostsim, inputs, lenient)


@pytest.mark.parametrize("seed", range(16))
def test_role_policy_chains_with_five_to_eight_roles(seed):
    rng = np.random.default_rng(89_000 + seed)
    docs, acts = _role_policy_store(rng)
    rt = rule_table_from_policies(policies_from_docs(docs))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    inputs = _more_roles(rng, _role_policy_requests(rng, acts, 220), ["staff", "lead", "temp", "vendor", "intern"], share=0.6)
    batch = Flattener(lt).flatten(inputs)
    for lenient in (False, True):
        _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0))
        assert _both_kernels.walk_wide & 1
        _against_oracle(rt, lt, _hostsim, inputs, lenient)


def test_only_wide_requests_and_base_only_batches_take_one_walk_each():
    """A batch whose requests all have five to eight roles launches the wider walk alone beside an idle base walk; a batch
    without such requests never launches it; CBH_NO_WALK2_WIDE=1 sends them to the general walk as before - same words."""
    rng = np.random.default_rng(5)
    rt = rule_table_from_policies(policies_from_docs(workloads.c5_policies()))
    lt = lower_rule_table(rt)
    base = workloads.c5_requests(300, seed=3).to_inputs()
    batch = Flattener(lt).flatten(base)
    hostsim_api.check(lt, batch, NOW, capi.F_WANT_DERIVED_ROLES, device_order=True)
    assert hostsim_api.last_kind() == 2 and not hostsim_api.last_walk_wide()
    pool = sorted({r for i in base for r in i["principal"]["roles"]})
    wide = _more_roles(rng, base, pool, share=1.0)
    wide = [i for i in wide if len(i["principal"]["roles"]) <= 8]
    batch = Flattener(lt).flatten(wide)
    assert int(batch.req_u32[7].min()) >= 5 and int(batch.req_u32[7].max()) == 8
    new = _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES)
    assert _both_kernels.walk_wide == 1
    os.environ["CBH_NO_WALK2_WIDE"] = "1"
    try:
        old = hostsim_api.check(lt, batch, NOW, capi.F_WANT_DERIVED_ROLES, device_order=True)
        assert hostsim_api.last_kind() == 2 and not hostsim_api.last_walk_wide()
    finally:
        os.environ.pop("CBH_NO_WALK2_WIDE", None)
    for f in ("effect", "policy", "scope", "edr", "status"):
        assert np.array_equal(getattr(new, f), getattr(old, f)), f
    _against_oracle(rt, lt, _hostsim, wide)


# ---------------------------------------------------------------------------------------------------------- targeted
def _role_policy_store(rng):
    """Role policies whose roles inherit from each other (a slot's set holds several roles with policies at one scope),
    at two scopes, with conditions, globs in the allow lists and a wildcard resource."""
    docs = [{"apiVersion": API, "derivedRoles": {"name": "dr", "definitions": [
        {"name": "owner", "parentRoles": ["staff", "lead"], "condition": _expr("R.attr.owner == P.id")},
        {"name": "any", "parentRoles": ["*"]}]}}]
    acts = ["view", "view:public", "edit", "edit:draft", "delete", "approve", "share", "export"]
    for kind in ("doc", "sheet"):
        for scope in ("", "acme", "acme.hr"):
            rules = []
            for _ in range(int(rng.integers(2, 6))):
                rule = {"actions": [str(a) for a in rng.choice(acts + ["view:*", "*"], size=int(rng.integers(1, 4)), replace=False)],
                        "effect": "EFFECT_ALLOW" if rng.random() < 0.75 else "EFFECT_DENY"}
                if rng.random() < 0.3:
                    rule["derivedRoles"] = [str(rng.choice(["owner", "any"]))]
                else:
                    rule["roles"] = [str(r) for r in rng.choice(["staff", "lead", "temp", "vendor", "intern", "*"], size=int(rng.integers(1, 3)), replace=False)]
                if rng.random() < 0.5:
                    rule["condition"] = _expr(str(rng.choice(["R.attr.public == true", "R.attr.amount > 50", "R.attr.missing == 1",
                                                              "P.attr.teams.exists(t, t == \"core\")", "R.attr.tags.region == \"eu\""])))
                rules.append(rule)
            pol = {"resource": kind, "version": "default", "rules": rules, "importDerivedRoles": ["dr"]}
            if scope:
                pol["scope"] = scope
                if rng.random() < 0.4:
                    pol["scopePermissions"] = "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS"
            docs.append({"apiVersion": API, "resourcePolicy": pol})
    chain = [("temp", ["staff"]), ("vendor", ["temp"]), ("intern", ["vendor", "staff"]), ("staff", [])]
    for role, parents in chain:
        for scope in ("", "acme"):
            if rng.random() < 0.25:
                continue
            rules = []
            for res in rng.choice(["doc", "sheet", "*"], size=int(rng.integers(1, 3)), replace=False):
                rule = {"resource": str(res), "allowActions": [str(a) for a in rng.choice(acts + ["view:*", "edit:*"], size=int(rng.integers(1, 4)), replace=False)]}
                if rng.random() < 0.4:
                    rule["condition"] = _expr(str(rng.choice(["R.attr.public == true", "R.attr.amount > 50", "R.attr.missing == 1"])))
                rules.append(rule)
            rp = {"role": role, "rules": rules}
            if parents:
                rp["parentRoles"] = parents
            if scope:
                rp["scope"] = scope
            docs.append({"apiVersion": API, "rolePolicy": rp})
    return docs, acts


def _role_policy_requests(rng, acts, n):
    out = []
    for i in range(n):
        na = int(rng.choice([1, 2, 4, 5, 8, 9, 12], p=[0.15, 0.15, 0.3, 0.15, 0.15, 0.05, 0.05]))
        actions = [str(a) for a in rng.choice(acts + ["view:x", "edit:y", "other"], size=min(na, 11), replace=False)]
        actions += ["extra%d" % k for k in range(na - len(actions))]
        nr = int(rng.choice([0, 1, 2, 3, 4, 5], p=[0.05, 0.35, 0.3, 0.15, 0.1, 0.05]))
        r_attr = {"owner": "p%d" % rng.integers(0, 3), "public": bool(rng.random() < 0.5), "amount": float(rng.integers(0, 100)),
                  "tags": {"region": str(rng.choice(["eu", "us"]))}}
        if rng.random() < 0.1:
            del r_attr["public"]
        inp = {"requestId": "q%d" % i, "actions": actions,
               "principal": {"id": "p%d" % rng.integers(0, 3), "roles": [str(r) for r in rng.choice(["staff", "lead", "temp", "vendor", "intern", "nobody"], size=nr, replace=False)],
                             "attr": {"teams": [["core"], ["ops"], []][int(rng.integers(0, 3))]}},
               "resource": {"kind": str(rng.choice(["doc", "sheet", "other"])), "id": "r%d" % i, "attr": r_attr}}
        if rng.random() < 0.6:
            inp["resource"]["scope"] = str(rng.choice(["acme", "acme.hr", "acme.hr.uk", "zzz"]))
        out.append(inp)
    return out


@pytest.mark.parametrize("seed", range(24))
def test_role_policy_chains(seed):
    rng = np.random.default_rng(88_000 + seed)
    docs, acts = _role_policy_store(rng)
    rt = rule_table_from_policies(policies_from_docs(docs))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    assert lt.stats["walk2"], lt.stats["walk2_refused"]
    inputs = _role_policy_requests(rng, acts, 260)
    batch = Flattener(lt).flatten(inputs)
    assert int(batch.req_u32[9].max()) > 8     # wider requests ride along on the general walk's lanes
    for lenient in (False, True):
        _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0))
        _against_oracle(rt, lt, _hostsim, inputs, lenient)


def _principal_store(rng):
    docs = []
    acts = ["view", "edit", "delete", "view:public", "approve"]
    for kind in ("doc", "sheet"):
        for scope in ("", "acme"):
            rules = [{"actions": [str(a) for a in rng.choice(acts + ["*"], size=int(rng.integers(1, 3)), replace=False)],
                      "roles": [str(r) for r in rng.choice(["user", "admin", "*"], size=1)],
                      "effect": "EFFECT_ALLOW" if rng.random() < 0.7 else "EFFECT_DENY"} for _ in range(int(rng.integers(1, 4)))]
            if rng.random() < 0.5:
                rules[0]["condition"] = _expr("R.attr.public == true")
            pol = {"resource": kind, "version": str(rng.choice(["default", "v2"])), "rules": rules}
            if scope:
                pol["scope"] = scope
            docs.append({"apiVersion": API, "resourcePolicy": pol})
    for i in range(4):
        for scope in ("", "acme", "acme.hr"):
            if rng.random() < 0.5:
                continue
            rules = {}
            for _ in range(int(rng.integers(1, 3))):
                res = str(rng.choice(["doc", "sheet", "*", "do*"]))
                entries = {}
                for _ in range(int(rng.integers(1, 4))):
                    a = str(rng.choice(acts + ["view:*", "*"]))
                    e = {"action": a, "effect": "EFFECT_ALLOW" if rng.random() < 0.5 else "EFFECT_DENY"}
                    if rng.random() < 0.4:
                        e["condition"] = _expr(str(rng.choice(["R.attr.public == true", "R.attr.missing == 1", "R.attr.amount > 50",
                                                                "R.attr.tags.region == \"eu\""])))
                    entries[a] = e
                rules[res] = {"resource": res, "actions": list(entries.values())}
            pp = {"principal": "p%d" % i, "version": str(rng.choice(["default", "v2"])), "rules": list(rules.values())}
            if scope:
                pp["scope"] = scope
                if rng.random() < 0.4:
                    pp["scopePermissions"] = "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS"
            docs.append({"apiVersion": API, "principalPolicy": pp})
    return docs, acts


@pytest.mark.parametrize("seed", range(24))
def test_principal_policies(seed):
    rng = np.random.default_rng(99_000 + seed)
    docs, acts = _principal_store(rng)
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    assert lt.stats["walk2"], lt.stats["walk2_refused"]
    inputs = []
    for i in range(240):
        inp = {"requestId": "q%d" % i, "actions": [str(a) for a in rng.choice(acts + ["view:x", "other"], size=int(rng.integers(0, 7)), replace=False)],
               "principal": {"id": "p%d" % rng.integers(0, 6), "roles": [str(r) for r in rng.choice(["user", "admin", "x"], size=int(rng.integers(0, 3)), replace=False)]},
               "resource": {"kind": str(rng.choice(["doc", "sheet", "dox", "other"])), "id": "r%d" % i,
                            "attr": {"public": bool(rng.random() < 0.5), "amount": float(rng.integers(0, 100)), "tags": {"region": "eu"}}}}
        if rng.random() < 0.5:
            inp["principal"]["scope"] = str(rng.choice(["acme", "acme.hr", "acme.hr.uk", "zz"]))
        if rng.random() < 0.5:
            inp["resource"]["scope"] = str(rng.choice(["acme", "acme.hr", "zz"]))
        if rng.random() < 0.4:
            inp["resource"]["policyVersion"] = "v2"
        if rng.random() < 0.4:
            inp["principal"]["policyVersion"] = "v2"
        inputs.append(inp)
    batch = Flattener(lt).flatten(inputs)
    for lenient in (False, True):
        _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0))
        _against_oracle(rt, lt, _hostsim, inputs, lenient)


def test_c5_sample_walk2_vs_general_walk():
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
    batch = workloads.c5_requests(6000).to_batch(Flattener(lt))
    _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES)


def test_c5w_sample_wide_walk_vs_general_walk_vs_oracle():
    """bench.py --workload C5W: C5's table, principals with five to eight roles - every request on cbh_walk2_wide_kernel."""
    rt = rule_table_from_policies(policies_from_docs(workloads.c5_policies()))
    lt = lower_rule_table(rt)
    cr = workloads.c5_requests(4000, roles_per_request=(5, 8))
    batch = cr.to_batch(Flattener(lt))
    assert int(batch.req_u32[7].min()) == 5 and int(batch.req_u32[7].max()) == 8
    _both_kernels(lt, batch, capi.F_WANT_DERIVED_ROLES)
    assert _both_kernels.walk_wide == 1
    _against_oracle(rt, lt, _hostsim, cr.head(400).to_inputs())


# ---------------------------------------------------------------------------------------------------------- GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_role_policy_chains(seed):
    rng = np.random.default_rng(88_000 + seed)
    docs, acts = _role_policy_store(rng)
    rt = rule_table_from_policies(policies_from_docs(docs))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    inputs = _role_policy_requests(rng, acts, 260)
    for lenient in (False, True):
        _against_oracle(rt, lt, lambda l: HipEvaluator(l, Conf()), inputs, lenient)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_fuzz_stores(seed):
    rng = np.random.default_rng(77_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    inputs = _requests(rng, 220)
    for lenient in (False, True):
        _against_oracle(rt, lt, lambda l: HipEvaluator(l, Conf()), inputs, lenient)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_gpu_wider_requests(seed):
    """cbh_walk2_wide_kernel / cbh_walk2_awide_kernel on the device: the fuzz stores and the role-policy chains with requests of
    up to ten roles and eighteen actions, one-shot (cbh_check_batch) against the oracle and resident (cbh_check_resident: the
    plan must name the wider walks) word for word what the one-shot call returned."""
    from test_fuzz_parity import ACTIONS, ROLES
    rng = np.random.default_rng(83_000 + seed)
    if seed % 2:
        docs, acts = _role_policy_store(rng)
        inputs = _more_roles(rng, _role_policy_requests(rng, acts, 220), ["staff", "lead", "temp", "vendor", "intern"], share=0.4)
        inputs = _more_actions(rng, inputs, acts + ["view:x", "edit:y", "other"], share=0.4)
    else:
        docs = _policies(rng)
        inputs = _more_actions(rng, _more_roles(rng, _requests(rng, 200), ROLES + ["other"], share=0.35), list(ACTIONS))
    rt = rule_table_from_policies(policies_from_docs(docs))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    if not lt.stats["walk2"]:
        pytest.skip("table stays on the general walk: %s" % lt.stats["walk2_refused"])
    for lenient in (False, True):
        _against_oracle(rt, lt, lambda l: HipEvaluator(l, Conf()), inputs, lenient)
    table = capi.Table(lt.blob)
    try:
        batch = Flattener(lt).flatten(inputs)
        flags = capi.F_WANT_DERIVED_ROLES
        one = table.check(batch, now_ns=NOW, flags=flags)
        db = table.upload(batch)
        plan = table.plan(db, flags)
        assert "cbh_walk2_wide_kernel" in plan and "cbh_walk2_awide_kernel" in plan, plan
        table.launch(db, now_ns=NOW, flags=flags)
        table.synchronize()
        res = table.download(db)
        for f in ("effect", "policy", "scope", "status", "edr"):
            assert np.array_equal(getattr(res, f), getattr(one, f)), f
        db.close()
    finally:
        table.close()


# ---------------------------------------------------------------------------------------------------------- trace marks
def _traced_fraction(make_evaluator):
    """The reference's own test store has policy variables and rules with outputs: a kernel that cannot tell which inputs the
    trace pass has something for marks them all (CBH_ST_WANTS_TRACE).  The walk marks the inputs that absorbed an error,
    visited a rule with outputs or have a failing variable - and what engine.check(trace=True) then returns must still be
    what the reference returned, input by input."""
    from helpers import load_json
    lt = lower_rule_table(store_rule_table(), GLOBALS)
    assert lt.trace_has_variables and lt.trace_has_outputs and lt.stats["walk2"]
    ev = make_evaluator(lt)
    cases = [c for c in load_json("engine_cases.json") if not c.get("strict")]
    total = traced = needed = 0
    try:
        for case in cases:
            lenient = bool(case["lenient"])
            outs, bad, incomplete = ev.check(case["inputs"], now_ns=NOW, lenient_scope_search=lenient, allow_unsupported=True, trace=True)
            traced += ev.last_traced
            for i, (have, want) in enumerate(zip(outs, case["wantOutputs"])):
                if i in bad:
                    continue
                total += 1
                needed += bool(want.get("evaluationErrors") or want.get("outputs"))
                what = incomplete.get(i, ())
                if "errors" not in what:
                    assert have["evaluationErrors"] == (want.get("evaluationErrors") or []), (case["name"], i)
                if "outputs" not in what:
                    assert sorted(have["outputs"], key=lambda o: o["src"]) == sorted(want.get("outputs") or [], key=lambda o: o["src"]), (case["name"], i)
    finally:
        if not isinstance(ev, HostSimEvaluator):
            ev.close()
    return total, traced, needed


def test_only_marked_inputs_are_traced():
    total, traced, needed = _traced_fraction(lambda lt: HostSimEvaluator(lt, Conf(globals_=GLOBALS)))
    assert needed <= traced < 0.6 * total, (total, traced, needed)


@pytest.mark.gpu
def test_gpu_only_marked_inputs_are_traced():
    total, traced, needed = _traced_fraction(lambda lt: HipEvaluator(lt, Conf(globals_=GLOBALS)))
    assert needed <= traced < 0.6 * total, (total, traced, needed)
