"""Runnable policy sets -> rule table (rows + side tables).

The rule table is the artefact both the device lowering (``cerbos_amd.lower``) and the
CPU oracle (``oracle/``) consume; it mirrors ``runtimev1.RuleTable``
(``api/private/cerbos/runtime/v1/runtime.proto:375-444``) as plain dicts.

Row generation restates ``internal/ruletable/ruletable.go``:
  addPrincipalPolicy :136-241, addResourcePolicy :243-414, addRolePolicy :416-473,
  indexRules (scope maps, scope permissions, derived roles) :798-834,
  parent-role closure ``index/index.go:749-788``.

Row order: the reference iterates Go maps (actions, roles, principal resources) so its
binding order is unspecified among rows of one rule; here it is file order. Effects do not
depend on it (rows of one rule share condition and effect).
"""
from __future__ import annotations

from .. import namer
from ..policy.compile import (SP_OVERRIDE_PARENT, SP_REQUIRE_PARENTAL_CONSENT,
                              SP_UNSPECIFIED, compile_all)

KIND_RESOURCE = "RESOURCE"
KIND_PRINCIPAL = "PRINCIPAL"


def _params(p):
    return {"ordered_variables": list(p["ordered_variables"]), "constants": dict(p["constants"])}


def _consent_rewrite(row, raw_sp):
    """REQUIRE_PARENTAL_CONSENT + conditional ALLOW -> DENY if none(cond)
    (ruletable.go:223-234, :336-347, :393-404)."""
    if raw_sp == SP_REQUIRE_PARENTAL_CONSENT and row["effect"] == "ALLOW" and row["condition"] is not None:
        row["condition"] = ("none", (row["condition"],))
        row["effect"] = "DENY"


def _blank_row(**kw):
    row = {
        "origin_fqn": "", "resource": "", "role": "", "action": None, "allow_actions": None,
        "condition": None, "derived_role_condition": None, "effect": None, "scope": "",
        "scope_permissions": SP_UNSPECIFIED, "version": "", "origin_derived_role": "",
        "emit_output": None, "name": "", "principal": "", "params": None,
        "derived_role_params": None, "evaluation_key": None, "policy_kind": KIND_RESOURCE,
        "from_role_policy": False,
    }
    row.update(kw)
    return row


def _add_principal_policy(rt, p):
    sp = p["scope_permissions"] or SP_OVERRIDE_PARENT
    rt["meta"][p["fqn"]] = {"fqn": p["fqn"], "kind": "principal", "name": p["principal"], "version": p["version"]}
    rows = []
    if not p["resource_rules"]:
        rows.append(_blank_row(origin_fqn=p["fqn"], scope=p["scope"], scope_permissions=sp,
                               version=p["version"], principal=p["principal"],
                               policy_kind=KIND_PRINCIPAL, params=_params({"ordered_variables": [], "constants": {}})))
    for resource, action_rules in p["resource_rules"].items():
        for rule in action_rules:
            row = _blank_row(
                origin_fqn=p["fqn"], resource=namer.sanitize(resource), role="*", action=rule["action"],
                condition=rule["condition"], effect=rule["effect"], scope=p["scope"], scope_permissions=sp,
                version=p["version"], emit_output=rule["emit_output"], name=rule["name"],
                principal=p["principal"], params=_params(p),
                evaluation_key=(namer.PRINCIPAL_POLICIES_PREFIX, "", p["principal"], "", "", p["version"],
                                p["scope"], rule["name"], 0),
                policy_kind=KIND_PRINCIPAL,
            )
            _consent_rewrite(row, p["scope_permissions"])
            rows.append(row)
    return rows


def _add_resource_policy(rt, p):
    sres = namer.sanitize(p["resource"])
    sp = p["scope_permissions"] or SP_OVERRIDE_PARENT
    rt["meta"][p["fqn"]] = {"fqn": p["fqn"], "kind": "resource", "name": sres, "version": p["version"]}
    if p["derived_roles"]:
        rt["policy_derived_roles"][p["fqn"]] = p["derived_roles"]
    rows = []
    if not p["rules"]:
        rows.append(_blank_row(origin_fqn=p["fqn"], resource=sres, scope=p["scope"], scope_permissions=sp,
                               version=p["version"], policy_kind=KIND_RESOURCE,
                               params=_params({"ordered_variables": [], "constants": {}})))
    for rule in p["rules"]:
        common = dict(origin_fqn=p["fqn"], resource=sres, effect=rule["effect"], scope=p["scope"],
                      scope_permissions=sp, version=p["version"], emit_output=rule["emit_output"],
                      name=rule["name"], policy_kind=KIND_RESOURCE)
        for a in rule["actions"]:
            for r in rule["roles"]:
                row = _blank_row(role=r, action=a, condition=rule["condition"], params=_params(p),
                                 evaluation_key=(namer.RESOURCE_POLICIES_PREFIX, sres, "", "", "", p["version"],
                                                 p["scope"], rule["name"], 0),
                                 **common)
                _consent_rewrite(row, p["scope_permissions"])
                rows.append(row)
            for dr in rule["derived_roles"]:
                rdr = p["derived_roles"].get(dr)
                if rdr is None:
                    continue
                for pr in rdr["parent_roles"]:
                    row = _blank_row(role=pr, action=a, condition=rule["condition"],
                                     derived_role_condition=rdr["condition"], origin_derived_role=dr,
                                     params=_params(p), derived_role_params=_params(rdr),
                                     evaluation_key=(namer.DERIVED_ROLES_PREFIX, sres, "", "", dr, p["version"],
                                                     p["scope"], rule["name"], 0),
                                     **common)
                    _consent_rewrite(row, p["scope_permissions"])
                    rows.append(row)
    return rows


def _add_role_policy(rt, p):
    rt["meta"][p["fqn"]] = {"fqn": p["fqn"], "kind": "role", "name": p["role"], "version": p["version"]}
    rows = []
    for resource, rules in p["resources"].items():
        for idx, rule in enumerate(rules):
            rows.append(_blank_row(
                origin_fqn=p["fqn"], role=p["role"], resource=resource, allow_actions=list(rule["allow_actions"]),
                condition=rule["condition"], emit_output=rule["emit_output"], name=rule["name"], scope=p["scope"],
                version=p["version"], params=_params(p),
                # idx restarts per resource and the resource is not part of the key (ruletable.go:445-455)
                evaluation_key=(namer.ROLE_POLICIES_PREFIX, "", "", p["role"], "", p["version"], p["scope"], "", idx),
                policy_kind=KIND_RESOURCE, from_role_policy=True,
            ))
    rt["scope_parent_roles"].setdefault(p["scope"], {})[p["role"]] = list(p["parent_roles"])
    return rows


def _compile_parent_role_ancestors(scope_parent_roles):
    """index/index.go:749-788 - transitive closure per scope. Ancestors are returned in
    sorted order (the reference's order comes from Go map iteration)."""
    compiled = {}
    for scope, roles in scope_parent_roles.items():
        compiled[scope] = {}
        for role in roles:
            seen, acc = set(), set()

            def collect(r):
                if r in seen:
                    return
                seen.add(r)
                for pr in roles.get(r, ()):
                    acc.add(pr)
                    collect(pr)

            collect(role)
            compiled[scope][role] = sorted(acc)
    return compiled


def build_rule_table(runnable_sets) -> dict:
    rt = {
        "rules": [],
        "meta": {},
        "scope_parent_roles": {},
        "policy_derived_roles": {},
    }
    for rps in runnable_sets:
        if rps["kind"] == "resource":
            rows = _add_resource_policy(rt, rps)
        elif rps["kind"] == "principal":
            rows = _add_principal_policy(rt, rps)
        else:
            rows = _add_role_policy(rt, rps)
        rt["rules"].extend(rows)
    for i, row in enumerate(rt["rules"]):
        row["id"] = i
    return finish_rule_table(rt)


def finish_rule_table(rt: dict) -> dict:
    """The side tables ``indexRules`` derives from the rows at load time (ruletable.go:798-834) - a table decoded
    from ``runtimev1.RuleTable`` bytes (``cerbos_amd.ruletable.proto``) gets them the same way."""
    rt["principal_scopes"] = sorted({r["scope"] for r in rt["rules"] if r["policy_kind"] == KIND_PRINCIPAL})
    rt["resource_scopes"] = sorted({r["scope"] for r in rt["rules"] if r["policy_kind"] == KIND_RESOURCE})
    sp = {}
    for r in rt["rules"]:
        if r["scope_permissions"] != SP_UNSPECIFIED:
            sp[r["scope"]] = r["scope_permissions"]  # last writer wins
    rt["scope_permissions"] = sp
    rt["parent_roles"] = _compile_parent_role_ancestors(rt["scope_parent_roles"])
    return rt


def rule_table_from_policies(policies: dict, sources: dict | None = None, require_ancestors: bool = False) -> dict:
    """`require_ancestors`: see policy.compile.compile_all - the entry points that read a policy directory pass True."""
    return build_rule_table(compile_all(policies, sources, require_ancestors))
