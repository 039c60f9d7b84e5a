"""Synthetic policy sets + request batches for the BASELINE.json configurations.

Seeds and shapes follow BASELINE.md "Synthetic configs" / SURVEY.md §8(d):
  C1  RBAC-only policy family of internal/engine/testdata/policy_template.yaml.gotmpl
      (10 rules x 12 literal actions, one role each, no CEL), 10k requests x 2 actions, seed 1
  C2  one resource policy `doc`, 5 conditional rules (4 ALLOW + 1 DENY), 1M tuples =
      250k requests x 4 actions, 2 % of requests miss an attribute, seed 2
  C3  10 kinds x 20 rules, scopes root/acme/acme.hr/acme.hr.uk (30 % REQUIRE_PARENTAL_CONSENT),
      2 derived-role sets, 1-3 of 12 roles per principal, 4M tuples, seed 3
All data is synthetic; generators are numpy-vectorised (``numpy.random.default_rng(seed)``).
"""
from __future__ import annotations

import numpy as np

from .columnar import Attr, ColumnarRequests, Ragged, Vocab

API = "api.cerbos.dev/v1"


def _expr(e):
    return {"match": {"expr": e}}


# ------------------------------------------------------------------------------------- C1
C1_ACTIONS = ["Create", "View", "Edit", "Delete", "Comment", "Flag", "Approve", "Reject", "Share", "Archive",
              "Restore", "Export"]


def c1_policies(n_sets=1):
    """policy_template.yaml.gotmpl rendered for N = 0..n_sets-1: resource kind
    ``resource_<N>``; rule i grants ``<Action><Suffix_i>`` actions to role ``role_<N>_<i>``."""
    suffixes = ["Reports", "Invoices", "Orders", "Users", "Teams", "Projects", "Tasks", "Files", "Notes", "Alerts"]
    docs = []
    for s in range(n_sets):
        rules = []
        for i, suf in enumerate(suffixes):
            rules.append({"actions": [a + suf for a in C1_ACTIONS], "effect": "EFFECT_ALLOW",
                          "roles": ["role_%d_%d" % (s, i)]})
        docs.append({"apiVersion": API, "resourcePolicy": {"resource": "resource_%d" % s, "version": "default",
                                                            "rules": rules}})
    return docs


def c1_requests(n_requests=10_000, seed=1, n_sets=1):
    rng = np.random.default_rng(seed)
    roles_v = ["role_%d_%d" % (s, i) for s in range(n_sets) for i in range(10)] + ["unknown_role"]
    ridx = rng.integers(0, n_sets * 10, n_requests)
    ridx = np.where(rng.random(n_requests) < 0.10, len(roles_v) - 1, ridx)
    kinds_v = ["resource_%d" % s for s in range(n_sets)]
    kidx = rng.integers(0, n_sets, n_requests)
    acts_v = ["ViewReports", "DeleteReports"]
    return ColumnarRequests(
        n_requests,
        principal_id=Vocab(["user_%d" % i for i in range(1000)], rng.integers(0, 1000, n_requests)),
        roles=Ragged(roles_v, np.arange(n_requests + 1), ridx),
        resource_kind=Vocab(kinds_v, kidx),
        resource_id=Vocab(["res_%d" % i for i in range(n_requests)], np.arange(n_requests)),
        actions=Ragged(acts_v, np.arange(n_requests + 1) * 2, np.tile([0, 1], n_requests)),
    )


# ------------------------------------------------------------------------------------- C2
C2_ACTIONS = ["view", "edit", "approve", "delete", "comment"]
C2_STATUS = ["OPEN", "PENDING", "CLOSED", "ARCHIVED"]
C2_DEPTS = ["eng", "ops", "sales", "legal", "hr", "finance", "support", "design"]


def c2_policies():
    rp = {
        "resource": "doc", "version": "default",
        "rules": [
            {"name": "public-view", "actions": ["view"], "roles": ["user", "manager"], "effect": "EFFECT_ALLOW",
             "condition": _expr("R.attr.public == true")},
            {"name": "owner", "actions": ["view", "edit"], "roles": ["user"], "effect": "EFFECT_ALLOW",
             "condition": _expr("R.attr.owner == P.id")},
            {"name": "big-approve", "actions": ["approve"], "roles": ["manager"], "effect": "EFFECT_ALLOW",
             "condition": _expr("R.attr.amount > 1000")},
            {"name": "same-dept", "actions": ["view", "comment"], "roles": ["user", "manager"], "effect": "EFFECT_ALLOW",
             "condition": _expr("P.attr.department == R.attr.department")},
            {"name": "frozen", "actions": ["edit", "delete"], "roles": ["user", "manager"], "effect": "EFFECT_DENY",
             "condition": _expr('R.attr.status in ["CLOSED", "ARCHIVED"]')},
            ],
    }
    return [{"apiVersion": API, "resourcePolicy": rp}]


def c2_requests(n_requests=250_000, seed=2, actions_per_request=4):
    rng = np.random.default_rng(seed)
    n = n_requests
    n_ids = 1000
    ids_v = ["u%04d" % i for i in range(n_ids)]
    pid = rng.integers(0, n_ids, n)
    owner = np.where(rng.random(n) < 0.10, pid, rng.integers(0, n_ids, n))
    role_choice = rng.integers(0, 3, n)  # 0 user, 1 manager, 2 both
    roles_v = ["user", "manager"]
    cnt = np.where(role_choice == 2, 2, 1)
    off = np.concatenate([[0], np.cumsum(cnt)])
    flat = np.zeros(off[-1], dtype=np.int64)
    flat[off[:-1]] = np.where(role_choice == 1, 1, 0)
    both = role_choice == 2
    flat[off[:-1][both] + 1] = 1
    # 2 % of requests drop one attribute (exercises the CEL error path)
    drop = rng.random(n) < 0.02
    which = rng.integers(0, 5, n)
    pres = [~(drop & (which == k)) for k in range(5)]
    # every request asks for `actions_per_request` distinct actions
    perm = np.argsort(rng.random((n, len(C2_ACTIONS))), axis=1)[:, :actions_per_request]
    return ColumnarRequests(
        n,
        principal_id=Vocab(ids_v, pid),
        roles=Ragged(roles_v, off, flat),
        resource_kind=Vocab(["doc"], np.zeros(n, dtype=np.int64)),
        resource_id=Vocab(["d%07d" % i for i in range(n)], np.arange(n)),
        actions=Ragged(C2_ACTIONS, np.arange(n + 1) * actions_per_request, perm.reshape(-1)),
        p_attr={"department": Attr("str", rng.integers(0, 8, n), pres[2], C2_DEPTS)},
        r_attr={
            "owner": Attr("str", owner, pres[0], ids_v),
            "amount": Attr("num", rng.random(n) * 2000.0, pres[1]),
            "department": Attr("str", rng.integers(0, 8, n), None, C2_DEPTS),
            "public": Attr("bool", rng.random(n) < 0.3, pres[3]),
            "status": Attr("str", rng.integers(0, 4, n), pres[4], C2_STATUS),
        },
    )


# ------------------------------------------------------------------------------------- C3
C3_SCOPES = ["", "acme", "acme.hr", "acme.hr.uk"]
C3_ROLES = ["role%02d" % i for i in range(12)]
C3_ACTIONS = ["view", "edit", "delete", "share", "approve", "create", "comment", "export"]


def c3_policies(seed=3, n_kinds=10, rules_per_kind=20):
    """10 kinds x 20 rules spread over the scope chain root..acme.hr.uk (5 rules per scope),
    30 % of the scoped policies REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS, two derived-role sets."""
    rng = np.random.default_rng(seed)
    docs = [
        {"apiVersion": API, "derivedRoles": {"name": "ownership", "definitions": [
            {"name": "owner", "parentRoles": C3_ROLES[:6], "condition": _expr("R.attr.owner == P.id")},
            {"name": "teammate", "parentRoles": ["*"], "condition": _expr("R.attr.team == P.attr.team")}]}},
        {"apiVersion": API, "derivedRoles": {"name": "org", "definitions": [
            {"name": "same_department", "parentRoles": C3_ROLES[3:],
             "condition": _expr("P.attr.department == R.attr.department")},
            {"name": "senior", "parentRoles": C3_ROLES, "condition": _expr("P.attr.level >= 5")}]}},
    ]
    conds = [None, None, "R.attr.public == true", "R.attr.amount > 500", "P.attr.level >= 3",
             'R.attr.status in ["OPEN", "PENDING"]', "R.attr.owner == P.id"]
    drs = ["owner", "teammate", "same_department", "senior"]
    per_scope = rules_per_kind // len(C3_SCOPES)
    for k in range(n_kinds):
        for scope in C3_SCOPES:
            rules = []
            for i in range(per_scope):
                rule = {"actions": [str(a) for a in rng.choice(C3_ACTIONS, size=int(rng.integers(1, 4)), replace=False)],
                        "effect": "EFFECT_DENY" if rng.random() < 0.2 else "EFFECT_ALLOW"}
                if rng.random() < 0.35:
                    rule["derivedRoles"] = [str(x) for x in rng.choice(drs, size=int(rng.integers(1, 3)), replace=False)]
                else:
                    rule["roles"] = [str(x) for x in rng.choice(C3_ROLES, size=int(rng.integers(1, 4)), replace=False)]
                c = conds[int(rng.integers(0, len(conds)))]
                if c:
                    rule["condition"] = _expr(c)
                rules.append(rule)
            rp = {"resource": "kind%02d" % k, "version": "default", "rules": rules,
                  "importDerivedRoles": ["ownership", "org"]}
            if scope:
                rp["scope"] = scope
                if rng.random() < 0.30:
                    rp["scopePermissions"] = "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS"
            docs.append({"apiVersion": API, "resourcePolicy": rp})
    return docs


def c3_requests(n_requests=1_000_000, seed=3, actions_per_request=4, n_kinds=10):
    rng = np.random.default_rng(seed + 1000)
    n = n_requests
    n_ids = 2000
    ids_v = ["p%04d" % i for i in range(n_ids)]
    pid = rng.integers(0, n_ids, n)
    owner = np.where(rng.random(n) < 0.15, pid, rng.integers(0, n_ids, n))
    cnt = rng.integers(1, 4, n)
    off = np.concatenate([[0], np.cumsum(cnt)])
    # distinct roles per principal: take a random rotation of the role list
    start = rng.integers(0, len(C3_ROLES), n)
    flat = (np.repeat(start, cnt) + (np.arange(off[-1]) - np.repeat(off[:-1], cnt)) * 5) % len(C3_ROLES)
    scopes_v = C3_SCOPES + ["acme.hr.uk.london", "other"]
    perm = np.argsort(rng.random((n, len(C3_ACTIONS))), axis=1)[:, :actions_per_request]
    teams = ["t%d" % i for i in range(16)]
    return ColumnarRequests(
        n,
        principal_id=Vocab(ids_v, pid),
        roles=Ragged(C3_ROLES, off, flat),
        resource_kind=Vocab(["kind%02d" % k for k in range(n_kinds)] + ["unknown_kind"],
                            np.where(rng.random(n) < 0.02, n_kinds, rng.integers(0, n_kinds, n))),
        resource_id=Vocab(["r%07d" % i for i in range(n)], np.arange(n)),
        actions=Ragged(C3_ACTIONS, np.arange(n + 1) * actions_per_request, perm.reshape(-1)),
        resource_scope=Vocab(scopes_v, rng.choice(len(scopes_v), n, p=[0.2, 0.2, 0.2, 0.3, 0.05, 0.05])),
        p_attr={"department": Attr("str", rng.integers(0, 8, n), None, C2_DEPTS),
                "team": Attr("str", rng.integers(0, 16, n), None, teams),
                "level": Attr("num", rng.integers(1, 9, n).astype(np.float64), rng.random(n) > 0.01)},
        r_attr={"owner": Attr("str", owner, None, ids_v),
                "team": Attr("str", rng.integers(0, 16, n), None, teams),
                "department": Attr("str", rng.integers(0, 8, n), None, C2_DEPTS),
                "amount": Attr("num", rng.random(n) * 1000.0, rng.random(n) > 0.01),
                "public": Attr("bool", rng.random(n) < 0.3),
                "status": Attr("str", rng.integers(0, 4, n), None, C2_STATUS)},
    )


# ------------------------------------------------------------------------------------- C4
C4_SCOPES = ["", "acme", "acme.hr"]
C4_ROLES = ["r%02d" % i for i in range(24)]
C4_ACTIONS = ["act%02d" % i for i in range(16)]
C4_REGIONS = ["eu", "us", "apac", "latam"]


def c4_condition_pool():
    """64 condition shapes over nine attributes: single comparisons and all/any/none trees of them."""
    leaves = (["R.attr.amount > %d" % v for v in (100, 250, 500, 750, 900)]
              + ["R.attr.amount <= %d" % v for v in (200, 400, 600, 800)]
              + ["P.attr.level >= %d" % v for v in range(1, 9)]
              + ['R.attr.region == "%s"' % r for r in C4_REGIONS]
              + ['R.attr.region != "%s"' % r for r in C4_REGIONS]
              + ["R.attr.owner == P.id", "R.attr.owner != P.id", "R.attr.public == true", "R.attr.public != true",
                 "P.attr.department == R.attr.department", "P.attr.department != R.attr.department",
                 "R.attr.team == P.attr.team", 'R.attr.status in ["OPEN", "PENDING"]',
                 'R.attr.status in ["CLOSED"]', 'R.attr.region in ["eu", "us"]'])
    pool = [_expr(e) for e in leaves]
    rng = np.random.default_rng(4040)
    kinds = ["all", "any", "none"]
    while len(pool) < 64:
        k = kinds[len(pool) % 3]
        picks = rng.choice(len(leaves), size=int(rng.integers(2, 4)), replace=False)
        pool.append({"match": {k: {"of": [{"expr": leaves[int(i)]} for i in picks]}}})
    return pool[:64]


def c4_policies(seed=4, n_policies=1000, rules_per_policy=50):
    """`n_policies` resource policies = kinds k0000.. x the scopes root/acme/acme.hr (three policies per
    kind), `rules_per_policy` rules each (1000 x 50 = 50k rules), 40 % with a condition of the pool."""
    rng = np.random.default_rng(seed)
    pool = c4_condition_pool()
    docs = []
    for p in range(n_policies):
        kind, scope = "k%04d" % (p // 3), C4_SCOPES[p % 3]
        rules = []
        for _ in range(rules_per_policy):
            rule = {"actions": [str(a) for a in rng.choice(C4_ACTIONS, size=int(rng.integers(1, 4)), replace=False)],
                    "roles": [str(r) for r in rng.choice(C4_ROLES, size=int(rng.integers(1, 4)), replace=False)],
                    "effect": "EFFECT_DENY" if rng.random() < 0.15 else "EFFECT_ALLOW"}
            if rng.random() < 0.40:
                rule["condition"] = pool[int(rng.integers(0, len(pool)))]
            rules.append(rule)
        rp = {"resource": kind, "version": "default", "rules": rules}
        if scope:
            rp["scope"] = scope
            if rng.random() < 0.25:
                rp["scopePermissions"] = "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS"
        docs.append({"apiVersion": API, "resourcePolicy": rp})
    return docs


def c4_requests(n_requests=500_000, seed=4, actions_per_request=4, n_policies=1000):
    """Zipf(1.1) over the kinds; 2M tuples per GPU at the default size (16M over 8 GPUs)."""
    rng = np.random.default_rng(seed + 2000)
    n = n_requests
    n_kinds = (n_policies + 2) // 3
    kidx = np.minimum(rng.zipf(1.1, n) - 1, n_kinds * 4) % (n_kinds + 1)      # the last index = an unknown kind
    n_ids = 4000
    ids_v = ["p%04d" % i for i in range(n_ids)]
    pid = rng.integers(0, n_ids, n)
    cnt = rng.integers(1, 4, n)
    off = np.concatenate([[0], np.cumsum(cnt)])
    start = rng.integers(0, len(C4_ROLES), n)
    flat = (np.repeat(start, cnt) + (np.arange(off[-1]) - np.repeat(off[:-1], cnt)) * 7) % len(C4_ROLES)
    scopes_v = C4_SCOPES + ["acme.hr.uk", "other"]
    perm = np.argsort(rng.random((n, len(C4_ACTIONS))), axis=1)[:, :actions_per_request]
    teams = ["t%d" % i for i in range(16)]
    return ColumnarRequests(
        n,
        principal_id=Vocab(ids_v, pid),
        roles=Ragged(C4_ROLES, off, flat),
        resource_kind=Vocab(["k%04d" % k for k in range(n_kinds)] + ["unknown_kind"], kidx),
        resource_id=Vocab(["r%07d" % i for i in range(n)], np.arange(n)),
        actions=Ragged(C4_ACTIONS, np.arange(n + 1) * actions_per_request, perm.reshape(-1)),
        resource_scope=Vocab(scopes_v, rng.choice(len(scopes_v), n, p=[0.25, 0.25, 0.3, 0.15, 0.05])),
        p_attr={"department": Attr("str", rng.integers(0, 8, n), None, C2_DEPTS),
                "team": Attr("str", rng.integers(0, 16, n), None, teams),
                "level": Attr("num", rng.integers(1, 9, n).astype(np.float64), rng.random(n) > 0.01)},
        r_attr={"owner": Attr("str", np.where(rng.random(n) < 0.15, pid, rng.integers(0, n_ids, n)), None, ids_v),
                "team": Attr("str", rng.integers(0, 16, n), None, teams),
                "department": Attr("str", rng.integers(0, 8, n), None, C2_DEPTS),
                "amount": Attr("num", rng.random(n) * 1000.0, rng.random(n) > 0.01),
                "public": Attr("bool", rng.random(n) < 0.3),
                "region": Attr("str", rng.integers(0, 4, n), rng.random(n) > 0.01, C4_REGIONS),
                "status": Attr("str", rng.integers(0, 4, n), None, C2_STATUS)},
    )


# ------------------------------------------------------------------------------------- T
def t_policies(seed=7):
    """north_star's target set: 100 resource policies x 100 rules = 10k rules, 40 % with a CEL condition of C4's pool
    (34 kinds x the scopes root / acme / acme.hr: 100 rules per (kind, scope) bucket)."""
    return c4_policies(seed=seed, n_policies=100, rules_per_policy=100)


def t_requests(n_requests=250_000, seed=7):
    return c4_requests(n_requests, seed=seed, n_policies=100)


# ------------------------------------------------------------------------------------- C5
C5_ACTIONS = ["view", "view:public", "view:internal", "edit", "edit:public", "delete", "share:public", "approve"]


def c5_policies(seed=5, n_principal_policies=100):
    """C3 plus: principal-policy overrides for 100 principals, action globs (``view:*``, ``*:public``,
    ``*``), role policies with allow-lists, and conditions over nested attributes (map / list values,
    comprehension macros) that need the operand-stack interpreter."""
    rng = np.random.default_rng(seed)
    docs = c3_policies(seed=3)
    for k in range(10):   # glob rules + nested conditions on every kind, root scope
        docs.append({"apiVersion": API, "resourcePolicy": {
            "resource": "kind%02d" % k, "version": "v5", "rules": [
                {"actions": ["view:*"], "roles": ["role00", "role01", "role02"], "effect": "EFFECT_ALLOW",
                 "condition": _expr("R.attr.tags.region in P.attr.tags.regions")},
                {"actions": ["*:public"], "roles": ["*"], "effect": "EFFECT_ALLOW"},
                {"actions": ["edit", "edit:public"], "roles": ["role03", "role04"], "effect": "EFFECT_ALLOW",
                 "condition": _expr('P.attr.teams.exists(t, t.startsWith("comm"))')},
                {"actions": ["*"], "roles": ["role05"], "effect": "EFFECT_ALLOW",
                 "condition": _expr("R.attr.acl[P.id].level >= 2")},
                {"actions": ["delete"], "roles": ["*"], "effect": "EFFECT_DENY",
                 "condition": _expr('"legal-hold" in R.attr.labels')},
            ]}})
    for i in range(n_principal_policies):   # principal overrides
        rules = [{"resource": "kind%02d" % int(rng.integers(0, 10)),
                  "actions": [{"action": str(rng.choice(["view", "edit", "delete", "view:*", "*"])),
                               "effect": "EFFECT_DENY" if rng.random() < 0.5 else "EFFECT_ALLOW"}
                              for _ in range(int(rng.integers(1, 3)))]}
                 for _ in range(int(rng.integers(1, 3)))]
        for r in rules:   # one action entry per action string
            seen = {}
            for a in r["actions"]:
                seen.setdefault(a["action"], a)
            r["actions"] = list(seen.values())
        merged = {}
        for r in rules:
            merged.setdefault(r["resource"], r)
        docs.append({"apiVersion": API, "principalPolicy": {"principal": "p%04d" % (i * 20), "version": "v5",
                                                             "rules": list(merged.values())}})
    for role, parents in (("contractor", ["role00"]), ("auditor", [])):   # role policies (allow-lists)
        rp = {"role": role, "rules": [
            {"resource": "kind00", "allowActions": ["view", "view:*"]},
            {"resource": "kind01", "allowActions": ["view:public", "share:public"],
             "condition": _expr("R.attr.public == true")},
            {"resource": "*", "allowActions": ["approve"]}]}
        if parents:
            rp["parentRoles"] = parents
        docs.append({"apiVersion": API, "rolePolicy": rp})
    return docs


def c5_requests(n_requests=250_000, seed=5, actions_per_request=4, roles_per_request=(1, 3)):
    """Mixed batch: half of the requests ask for the v5 policies (globs + nested conditions), 5 % of the
    principals have a principal policy, 10 % hold a role-policy role; 1M tuples per GPU by default.
    ``roles_per_request`` = (lowest, highest) number of roles of a principal (at most 8: distinct names of C3_ROLES)."""
    rng = np.random.default_rng(seed + 3000)
    n = n_requests
    base = c3_requests(n, seed=seed, actions_per_request=actions_per_request)
    n_ids = 2000
    ids_v = ["p%04d" % i for i in range(n_ids)]
    pid = rng.integers(0, n_ids, n)
    roles_v = C3_ROLES + ["contractor", "auditor"]
    assert 1 <= roles_per_request[0] <= roles_per_request[1] <= 8
    cnt = rng.integers(roles_per_request[0], roles_per_request[1] + 1, n)
    off = np.concatenate([[0], np.cumsum(cnt)])
    start = rng.integers(0, len(C3_ROLES), n)
    flat = (np.repeat(start, cnt) + (np.arange(off[-1]) - np.repeat(off[:-1], cnt)) * 5) % len(C3_ROLES)
    special = rng.random(n) < 0.10
    flat[off[:-1][special]] = len(C3_ROLES) + rng.integers(0, 2, int(special.sum()))
    perm = np.argsort(rng.random((n, len(C5_ACTIONS))), axis=1)[:, :actions_per_request]
    regions = ["eu", "us", "apac"]
    tag_v = [{"region": r} for r in regions] + [{"zone": "z1"}]
    ptag_v = [{"regions": ["eu"]}, {"regions": ["eu", "us"]}, {"regions": []}, {"regions": ["apac", "us", "eu"]}]
    teams_v = [["commerce", "ops"], ["core"], [], ["community", "design", "comms"]]
    labels_v = [[], ["legal-hold"], ["pii", "legal-hold"], ["pii"]]
    acl_ids = ["p%04d" % i for i in range(0, n_ids, 40)]
    acl_v = [{}] + [{a: {"level": lvl}} for a in acl_ids[:25] for lvl in (1, 3)]
    return ColumnarRequests(
        n,
        principal_id=Vocab(ids_v, pid),
        roles=Ragged(roles_v, off, flat),
        resource_kind=base.resource_kind,
        resource_id=base.resource_id,
        actions=Ragged(C5_ACTIONS, np.arange(n + 1) * actions_per_request, perm.reshape(-1)),
        resource_scope=base.resource_scope,
        principal_version=Vocab(["default", "v5"], (rng.random(n) < 0.5).astype(np.int64)),
        resource_version=Vocab(["default", "v5"], (rng.random(n) < 0.5).astype(np.int64)),
        p_attr=dict(base.p_attr, tags=Attr("json", rng.integers(0, len(ptag_v), n), rng.random(n) > 0.02, ptag_v),
                    teams=Attr("json", rng.integers(0, len(teams_v), n), None, teams_v)),
        r_attr=dict(base.r_attr, tags=Attr("json", rng.integers(0, len(tag_v), n), None, tag_v),
                    labels=Attr("json", rng.integers(0, len(labels_v), n), rng.random(n) > 0.02, labels_v),
                    acl=Attr("json", rng.integers(0, len(acl_v), n), None, acl_v)),
    )


# ------------------------------------------------------------------------------------- hack/loadtest
# BASELINE.json configs[0] names the reference's own load-test sets.  tests/golden/loadtest_templates.json holds
# the template text of hack/loadtest/templates/{classic,multitenant} (tools/make_golden_loadtest.py); rendering
# follows hack/loadtest/generate.go:49-51,84-100: every template once per N in 0..count-1 with
# NameMod(x) = "%s_%05d" % (x, N) and RequestID = "REQ_%05d" % N, static files copied once.
_LOADTEST_FIXTURE = None


def _loadtest_fixture():
    global _LOADTEST_FIXTURE
    if _LOADTEST_FIXTURE is None:
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                            "loadtest_templates.json")
        with open(path, encoding="utf-8") as f:
            _LOADTEST_FIXTURE = json.load(f)
    return _LOADTEST_FIXTURE


def loadtest_policies(set_name="classic", count=10):
    """Policy documents of `./loadtest.sh -g` with NUM_POLICIES=count (schemas dropped: validation is out of scope)."""
    from .policy.loader import load_yaml_documents
    fx = _loadtest_fixture()[set_name]
    docs = []
    for t in fx["static_policies"]:
        docs.extend(load_yaml_documents(t["text"]))
    for n in range(count):
        for t in fx["policies"]:
            docs.extend(load_yaml_documents(t["text"].replace("@N@", "%05d" % n)))
    for d in docs:
        for k in ("resourcePolicy", "principalPolicy"):
            if k in d:
                d[k].pop("schemas", None)
    return docs


def loadtest_inputs(set_name="classic", count=10):
    """The sets' CheckResourcesRequest templates as CheckInputs, one per resource entry
    (svc/cerbos_svc.go:274-287), in template order for N = 0..count-1."""
    import json
    fx = _loadtest_fixture()[set_name]
    out = []
    for n in range(count):
        for t in fx["requests"]:
            req = json.loads(t["text"].replace("@N@", "%05d" % n))
            for k, entry in enumerate(req["resources"]):
                inp = {"requestId": "%s/%s/%d" % (req.get("requestId", ""), t["file"], k),
                       "principal": req["principal"], "resource": entry["resource"], "actions": entry["actions"]}
                if req.get("auxData"):
                    inp["auxData"] = req["auxData"]
                out.append(inp)
    return out


def c5w_requests(n_requests=250_000, seed=5):
    """C5's table, principals with five to eight roles (cbh_walk2_wide_kernel's shape)."""
    return c5_requests(n_requests, seed=seed, roles_per_request=(5, 8))
