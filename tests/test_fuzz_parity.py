"""CPU tier: differential fuzzing of the decision kernel's source (host-simulated waves, the real
lowering / flattening / response assembly) against oracle/check.py on randomly generated policy
sets and requests.

Every seed draws a small store that mixes the features of the path - scoped resource policies with
scope permissions, derived roles, principal policies, role policies with parent roles, literal and
glob actions / roles, `*`, conditions from single comparisons up to comprehensions over nested
attributes - and a batch of requests with ragged role / action lists (including more than 64 actions,
which the flattener splits), unknown kinds / scopes, missing and wrongly typed attributes.  Compared per
action: effect, policy key and scope; per request: effective derived roles.  A store the lowering
refuses (LoweringError: the reference's own result is history dependent there) is skipped, a request the
device flags UNSUPPORTED is skipped - neither may produce a wrong answer silently.

CBH_FUZZ_SEEDS=<n> widens the run (default 30 seeds).
"""
import os

import numpy as np
import pytest

from cerbos_amd.engine import Conf
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.lower.celc import LoweringError
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle
from test_hostsim_golden import HostSimEvaluator

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000
SCOPES = ["", "acme", "acme.hr"]
KINDS = ["doc", "album:photo", "report"]
ROLES = ["user", "manager", "admin", "guest", "contractor"]
ACTIONS = ["view", "view:public", "view:internal", "edit", "delete", "approve", "share:public"]
ACTION_PATTERNS = ACTIONS + ["*", "view:*", "*:public", "ed*"]
ROLE_PATTERNS = ROLES + ["*", "man*"]
DEPTS = ["eng", "ops", "legal"]
CONDITIONS = [
    "R.attr.public == true",
    "R.attr.owner == P.id",
    "R.attr.owner != P.id",
    "R.attr.amount > 100",
    "R.attr.amount <= 500",
    "P.attr.level >= 3",
    "P.attr.department == R.attr.department",
    'R.attr.status in ["OPEN", "PENDING"]',
    'R.attr.department in ["eng", "ops"]',
    'R.attr.tags.region in P.attr.regions',
    'P.attr.teams.exists(t, t.startsWith("co"))',
    "R.attr.acl[P.id] >= 2",
    'has(R.attr.tags.region) && R.attr.tags.region != "us"',
    "R.attr.amount > 100 && P.attr.level < 5 || R.attr.public == true",
    'size(P.attr.teams) > 1',
    '"admin" in P.roles',
    'R.kind == "doc" || R.id.startsWith("r1")',
]
# wider CEL surface (SURVEY.md §8a "device-subset CEL"): whatever the device cannot run must be flagged, never guessed
CONDITIONS_WIDE = CONDITIONS + [
    "R.attr.amount + 10.0 > 110.0",
    "R.attr.amount * 2.0 <= 900.0 || R.attr.amount / 4.0 > 150.0",
    "P.attr.level % 2.0 == 0.0" ,
    "!(R.attr.public == true) && P.attr.level > 1",
    'R.attr.public == true ? P.attr.level > 2 : R.attr.owner == P.id',
    'size(R.attr.status) == 4',
    'R.attr.status.endsWith("ING") || R.attr.status.contains("PE")',
    'P.id in R.attr.acl',
    'P.attr.teams.all(t, size(t) > 2)',
    'P.attr.teams.exists_one(t, t == "ops")',
    'P.attr.regions.exists(r, r == R.attr.tags.region)',
    'has(R.attr.acl) && size(R.attr.acl) > 0',
    'timestamp(R.attr.created) < now()',
    'now() - timestamp(R.attr.created) > duration("24h")',
    'timestamp(R.attr.created).timeSince() > duration("1h")',
    'R.attr.ip.inIPAddrRange("10.0.0.0/8")',
    'request.auxData.jwt.iss == "cerbos" && "admin" in request.auxData.jwt.groups',
    'request.aux_data.jwt.level > 2',
    '"eu" in P.attr.regions && !("us" in P.attr.regions)',
    'R.attr.amount > 100 == (P.attr.level > 3)',
    'P.attr.department != R.attr.department && R.attr.status != "CLOSED"',
    'R.attr.tags.region == "eu" || R.attr.tags.zone == "z"',
    'size(P.roles) > 1 && P.roles.exists(r, r.startsWith("man"))',
    'R.attr.owner.startsWith("p") && R.attr.owner.size() == 2',
    'R.scope == "acme" || P.scope == "acme.hr"',
    'R.policyVersion == "v2"',
    # lists built in the lane's arena (cbh_vm.h): filter / map / intersect / except / concatenation
    'size(P.attr.teams.filter(t, t.startsWith("co"))) > 0',
    '"ops" in P.attr.teams.filter(t, size(t) < 5)',
    'P.attr.regions.map(r, r == "eu").exists(b, b)',
    'P.attr.teams.map(t, size(t)).all(n, n > 3)',
    'intersect(P.attr.regions, ["eu", "apac"]) == ["eu"]',
    'P.attr.regions.except(["us"]).size() >= 1',
    'size(P.attr.regions + ["zz"]) > 2 || "zz" in (P.attr.teams + ["zz"])',
    'P.attr.teams.exists(t, P.attr.regions.filter(r, size(r) == size(t)).size() > 0)',
    'intersect(request.auxData.jwt.groups, P.attr.teams.map(t, t)).size() == 0',
    'P.attr.regions.filter(r, r != R.attr.tags.region) == P.attr.regions',
    'P.attr.teams.map(t, size(t) > 3, t).size() == 1',
    'P.attr.teams.transformList(i, t, i + size(t)).exists(n, n > 8)',
    'P.attr.regions.transformList(i, r, i > 0, r) == P.attr.regions.slice(1, size(P.attr.regions))',
    'lists.range(3).exists(i, i == size(P.attr.teams))',
    '(P.attr.teams + ["zz"]).reverse()[0] == "zz"',
    'P.attr.regions.slice(0, 2).size() == 2',
    # strings put together on the device are ropes (cbh_vm.h): never built, read part by part
    '(R.attr.department + ":" + R.attr.status) in ["eng:OPEN", "ops:PENDING", "legal:CLOSED"]',
    '(P.id + "@" + R.attr.department).endsWith("@eng") || (R.kind + "/" + R.id).startsWith("doc/r1")',
    'R.attr.status.lowerAscii() in ["open", "closed"] && R.attr.department.upperAscii() != "OPS"',
    'P.attr.teams.exists(t, (t + "-" + R.attr.department) == "ops-ops") || size("id:" + P.id) == 5',
    '("x" + R.attr.status).lowerAscii().contains("xo") && P.attr.teams.map(t, t + "!").exists(x, x == "core!")',
]


def _cond(rng, pool):
    r = rng.random()
    if r < 0.45:
        return {"match": {"expr": str(rng.choice(pool))}}
    kind = str(rng.choice(["all", "any", "none"]))
    return {"match": {kind: {"of": [{"expr": str(e)} for e in rng.choice(pool, size=int(rng.integers(2, 4)), replace=False)]}}}


def _policies(rng, wide=True):
    # half of the stores use single comparisons only (the leaf kernels), and principal / role policies
    # are left out often enough that every kernel feature class gets its share of seeds
    r = rng.random()
    if wide:
        pool = CONDITIONS[:10] if r < 0.4 else (CONDITIONS if r < 0.7 else CONDITIONS_WIDE)
    else:     # the generator as the GPU tier was validated with (tests/test_gpu_fuzz.py)
        pool = CONDITIONS[:10] if r < 0.5 else CONDITIONS
    with_principal, with_roles = rng.random() < 0.4, rng.random() < 0.4
    action_patterns = ACTION_PATTERNS if rng.random() < 0.6 else ACTIONS
    role_patterns = ROLE_PATTERNS if rng.random() < 0.6 else ROLES
    docs = [{"apiVersion": API, "derivedRoles": {"name": "common", "definitions": [
        {"name": "owner", "parentRoles": ["user", "manager"], "condition": {"match": {"expr": "R.attr.owner == P.id"}}},
        {"name": "anyone", "parentRoles": ["*"] if rng.random() < 0.6 else ["user", "guest", "admin"]},
        {"name": "senior", "parentRoles": [str(r) for r in rng.choice(ROLES, size=2, replace=False)],
         "condition": {"match": {"expr": "P.attr.level >= 4"}}}]}}]
    versions = ["default", "v2"]
    for kind in KINDS:
        for scope in SCOPES:
            if rng.random() < 0.25:
                continue
            for ver in versions[:1 + int(rng.random() < 0.3)]:
                rules = []
                for _ in range(int(rng.integers(1, 6))):
                    rule = {"actions": [str(a) for a in rng.choice(action_patterns, size=int(rng.integers(1, 4)), replace=False)],
                            "effect": "EFFECT_DENY" if rng.random() < 0.3 else "EFFECT_ALLOW"}
                    r = rng.random()
                    if r < 0.25:
                        rule["derivedRoles"] = [str(x) for x in rng.choice(["owner", "anyone", "senior"], size=int(rng.integers(1, 3)), replace=False)]
                    elif r < 0.35:
                        rule["roles"] = [str(x) for x in rng.choice(ROLES, size=1)]
                        rule["derivedRoles"] = ["owner"]
                    else:
                        rule["roles"] = [str(x) for x in rng.choice(role_patterns, size=int(rng.integers(1, 4)), replace=False)]
                    if rng.random() < 0.6:
                        rule["condition"] = _cond(rng, pool)
                    rules.append(rule)
                rp = {"resource": kind, "version": ver, "rules": rules, "importDerivedRoles": ["common"]}
                if scope:
                    rp["scope"] = scope
                    if rng.random() < 0.4:
                        rp["scopePermissions"] = "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS"
                docs.append({"apiVersion": API, "resourcePolicy": rp})
    for i in range(int(rng.integers(1, 3)) if with_principal else 0):   # principal policies
        rules = {}
        for _ in range(int(rng.integers(1, 3))):
            kind = str(rng.choice(KINDS + ["*"]))
            acts = {}
            for _ in range(int(rng.integers(1, 3))):
                a = str(rng.choice(ACTION_PATTERNS))
                e = {"action": a, "effect": "EFFECT_DENY" if rng.random() < 0.5 else "EFFECT_ALLOW"}
                if rng.random() < 0.4:
                    e["condition"] = {"match": {"expr": str(rng.choice(CONDITIONS[:9]))}}
                acts[a] = e
            rules[kind] = {"resource": kind, "actions": list(acts.values())}
        pp = {"principal": "p%d" % i, "version": "default", "rules": list(rules.values())}
        if rng.random() < 0.3:
            pp["scope"] = "acme"
        docs.append({"apiVersion": API, "principalPolicy": pp})
    if with_roles:   # role policies
        for role, parents in (("contractor", ["user"]), ("guest", [])):
            if rng.random() < 0.7:
                rules = [{"resource": str(rng.choice(KINDS + ["*"])),
                          "allowActions": [str(a) for a in rng.choice(ACTION_PATTERNS, size=int(rng.integers(1, 3)), replace=False)]}
                         for _ in range(int(rng.integers(1, 3)))]
                seen = {}
                for r in rules:
                    seen.setdefault(r["resource"], r)
                rules = list(seen.values())
                if rng.random() < 0.4:
                    rules[0]["condition"] = {"match": {"expr": str(rng.choice(CONDITIONS[:9]))}}
                rp = {"role": role, "rules": rules}
                if parents and rng.random() < 0.6:
                    rp["parentRoles"] = parents
                if rng.random() < 0.3:
                    rp["scope"] = "acme"
                docs.append({"apiVersion": API, "rolePolicy": rp})
    return docs


def _value(rng, kind):
    if kind == "num":
        return float(rng.integers(0, 800)) if rng.random() < 0.9 else "oops"
    if kind == "bool":
        return bool(rng.random() < 0.4) if rng.random() < 0.95 else "true"
    raise AssertionError(kind)


def _requests(rng, n, wide=True):
    out = []
    for i in range(n):
        pid = "p%d" % int(rng.integers(0, 6))
        p_attr, r_attr = {}, {}
        if rng.random() < 0.9:
            p_attr["level"] = _value(rng, "num") if rng.random() < 0.2 else float(rng.integers(1, 7))
        if rng.random() < 0.9:
            p_attr["department"] = str(rng.choice(DEPTS))
        if rng.random() < 0.8:
            p_attr["regions"] = [str(x) for x in rng.choice(["eu", "us", "apac"], size=int(rng.integers(0, 3)), replace=False)]
        if rng.random() < 0.8:
            p_attr["teams"] = [["core"], ["commerce", "ops"], [], ["community"]][int(rng.integers(0, 4))]
        if rng.random() < 0.9:
            r_attr["owner"] = pid if rng.random() < 0.4 else "p%d" % int(rng.integers(0, 6))
        if rng.random() < 0.9:
            r_attr["amount"] = _value(rng, "num")
        if rng.random() < 0.9:
            r_attr["public"] = _value(rng, "bool")
        if rng.random() < 0.9:
            r_attr["department"] = str(rng.choice(DEPTS))
        if rng.random() < 0.9:
            r_attr["status"] = str(rng.choice(["OPEN", "PENDING", "CLOSED"]))
        if rng.random() < 0.7:
            r_attr["tags"] = {"region": str(rng.choice(["eu", "us"]))} if rng.random() < 0.8 else {"zone": "z"}
        if rng.random() < 0.6:
            r_attr["acl"] = {("p%d" % int(rng.integers(0, 6))): float(rng.integers(1, 4)) for _ in range(int(rng.integers(0, 3)))}
        if wide and rng.random() < 0.8:
            r_attr["created"] = str(rng.choice(["2023-11-14T00:00:00Z", "2023-11-14T22:00:00Z", "2024-01-01T00:00:00Z", "yesterday"]))
        if wide and rng.random() < 0.8:
            r_attr["ip"] = str(rng.choice(["10.1.2.3", "192.168.0.1", "10.255.0.9", "not-an-ip"]))
        aux = None
        if wide and rng.random() < 0.6:
            aux = {"jwt": {"iss": str(rng.choice(["cerbos", "other"])), "groups": [str(x) for x in rng.choice(["admin", "dev", "ops"], size=int(rng.integers(0, 3)), replace=False)]}}
            if rng.random() < 0.7:
                aux["jwt"]["level"] = float(rng.integers(0, 6))
        n_act = int(rng.choice([1, 2, 3, 4, 5, 7, 40, 70], p=[0.2, 0.2, 0.2, 0.2, 0.1, 0.06, 0.02, 0.02]))
        if n_act <= len(ACTIONS):
            actions = [str(a) for a in rng.choice(ACTIONS, size=n_act, replace=False)]
        else:
            actions = ACTIONS + ["act%d" % k for k in range(n_act - len(ACTIONS))]
        inp = {"requestId": "q%d" % i,
               "principal": {"id": pid, "roles": [str(r) for r in rng.choice(ROLES + ["other"], size=int(rng.integers(0, 4)), replace=False)],
                             "attr": p_attr},
               "resource": {"kind": str(rng.choice(KINDS + ["unknown"])), "id": "r%d" % int(rng.integers(0, 30)), "attr": r_attr},
               "actions": actions}
        if aux is not None:
            inp["auxData"] = aux
        if rng.random() < 0.5:
            inp["resource"]["scope"] = str(rng.choice(SCOPES + ["acme.hr.uk", "zzz"]))
        if rng.random() < 0.3:
            inp["principal"]["scope"] = str(rng.choice(SCOPES + ["acme.hr.uk"]))
        if rng.random() < 0.2:
            inp["resource"]["policyVersion"] = "v2"
        out.append(inp)
    return out


SEEDS = list(range(int(os.environ.get("CBH_FUZZ_SEEDS", "30"))))


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_store(seed):
    rng = np.random.default_rng(10_000 + seed)
    docs = _policies(rng)
    rt = rule_table_from_policies(policies_from_docs(docs))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering (history-dependent reference behaviour)")
    ev = HostSimEvaluator(lt, Conf())
    orc = RuleTableOracle(rt)
    inputs = _requests(rng, 150)
    compared = 0
    for lenient, strict in ((False, False), (True, False), (False, True)):
        outs, bad = ev.check(inputs, now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict, allow_unsupported=True)
        params = EvalParams(now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict)
        for i, (inp, have) in enumerate(zip(inputs, outs)):
            if i in bad:
                continue
            want = orc.check(inp, params)
            assert norm_actions(have) == norm_actions(want), (seed, lenient, strict, inp)
            assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or []), (seed, inp)
            compared += 1
    assert compared > 150, "too many requests outside the device subset (%d compared)" % compared
