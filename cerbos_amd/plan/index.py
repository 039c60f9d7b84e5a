"""Index.Query over the rule table's rows for the query planner (internal/ruletable/index/index.go:214-336, 352-530): the bindings of
one (version, resource, scope, action, roles, policy kind, principal), role-policy synthetic DENYs first.  Host-side, symbolic path:
a scan with a (version, scope) pre-selection, not the device's tables (PlanResources is not the batched hot path)."""
from __future__ import annotations

from ..lower.globs import GlobNFA, fix_glob
from ..ruletable.build import KIND_PRINCIPAL, KIND_RESOURCE

_NFA = {}


def glob_match(pattern: str, value: str) -> bool:
    nfa = _NFA.get(pattern)
    if nfa is None:
        nfa = _NFA[pattern] = GlobNFA([fix_glob(pattern)])
    return bool(nfa.match_bits(value.encode("utf-8")) & 1)


def dim_match(key: str, value: str) -> bool:
    """A dimension key is a pattern only if it holds a '*' (index/glob_dimension.go:31-118)."""
    return glob_match(key, value) if "*" in key else key == value


def action_match(pattern: str, action: str) -> bool:
    """util.MatchesGlob as the allow-actions test uses it (index.go:446-449)."""
    return pattern == action or glob_match(pattern, action)


class PlanIndex:
    def __init__(self, rt: dict):
        self.rt = rt
        self.by_vs = {}
        for r in rt["rules"]:
            self.by_vs.setdefault((r["version"], r["scope"]), []).append(r)
        self.parent_roles = rt["parent_roles"]
        self.has_role_policies = any(r["allow_actions"] is not None for r in rt["rules"])

    def add_parent_roles(self, scopes, roles):
        """index.go:716-742: the roles followed by the parents the scopes' role policies give them."""
        merged = {}
        for s in scopes:
            for role, parents in (self.parent_roles.get(s) or {}).items():
                merged.setdefault(role, []).extend(parents)
        out = list(roles)
        for r in roles:
            out.extend(merged.get(r, ()))
        return out

    def query(self, version, resource, scope, action, roles, policy_kind, principal_id):
        rows = self.by_vs.get((version, scope), ())
        roles = list(roles)

        def base(r):   # every dimension but the action
            if r["resource"] and not dim_match(r["resource"], resource):
                return False
            if not r["resource"]:
                return False
            if roles and not any(dim_match(r["role"], x) for x in roles if r["role"]):
                return False
            if r["policy_kind"] != policy_kind:
                return False
            if principal_id != "" and r["principal"] != principal_id:
                return False
            return True
        based = [r for r in rows if base(r)]
        if not based:
            return []
        out = []
        if policy_kind == KIND_RESOURCE and self.has_role_policies:
            out.extend(self._role_policy_denies(rows, resource, roles, action, version, scope))
        for r in based:
            if r["allow_actions"] is None and r["action"] is not None and dim_match(r["action"], action):
                out.append(r)
        return out

    def _role_policy_denies(self, rows, resource, roles, action, version, scope):
        """index.go:352-530 for one resource and one action."""
        cand = [r for r in rows if r["allow_actions"] is not None and (not roles or any(dim_match(r["role"], x) for x in roles))]
        if not cand:
            return []
        rep, order = {}, []
        for r in cand:
            if r["role"] not in rep:
                rep[r["role"]] = r
                order.append(r["role"])
        by_role = {}
        for r in cand:
            if dim_match(r["resource"], resource):
                by_role.setdefault(r["role"], []).append(r)
        out = []
        for role in (roles or order):
            first = rep.get(role)
            if first is None:
                continue
            mine = by_role.get(role, [])
            if not mine:
                out.append(_no_match_deny(first, role, resource, action))
                continue
            matched = [r for r in mine if any(action_match(a, action) for a in r["allow_actions"])]
            if not matched:
                out.append(_no_match_deny(mine[0], role, mine[0]["resource"], action))
                continue
            for r in matched:
                if r["condition"] is None:
                    continue   # a plain allow falls through (outputs are not the planner's business)
                out.append(dict(r, effect="DENY", condition=("none", (r["condition"],)), action=action, allow_actions=None, from_role_policy=True))
        return out


def _no_match_deny(rep, role, resource, action):
    return dict(rep, effect="DENY", condition=None, action=action, allow_actions=None, from_role_policy=True, resource=resource, role=role,
                derived_role_condition=None, derived_role_params=None, params=None, no_match_for_scope_permissions=True)
