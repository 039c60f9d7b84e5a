"""Policy dicts -> runnable policy sets (the compile step that feeds the rule table).

Restates the parts of the reference's ``internal/compile`` the hot path depends on:
derived-role import resolution (``compile.go:252-327``), per-definition used
constants/variables (``compile.go:329-367``), rule auto-naming
(``compile.go:231-233``, ``namer_non_embedded.go:106-121``), the "used" variable set with
transitive references and stable topological order (``variables.go:236-305``) and used
constants (``constants.go:164-196``). No type checking is done here: expressions are
parsed (syntax errors surface) but not type-checked; the reference would reject some
ill-typed policies at compile time that this front-end lets through.

A condition is a tuple tree:  ('expr', text) | ('all'|'any'|'none', (cond, ...)).
"""
from __future__ import annotations

from .. import namer
from ..cel import parser as celparser
from .loader import policy_kind

ANY_ROLE = "*"

SP_UNSPECIFIED = 0
SP_OVERRIDE_PARENT = 1
SP_REQUIRE_PARENTAL_CONSENT = 2

_SP_NAMES = {
    None: SP_UNSPECIFIED,
    "": SP_UNSPECIFIED,
    "SCOPE_PERMISSIONS_UNSPECIFIED": SP_UNSPECIFIED,
    "SCOPE_PERMISSIONS_OVERRIDE_PARENT": SP_OVERRIDE_PARENT,
    "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS": SP_REQUIRE_PARENTAL_CONSENT,
}

_EFFECTS = {"EFFECT_ALLOW": "ALLOW", "EFFECT_DENY": "DENY"}


class CompileError(ValueError):
    pass


def compile_condition(cond):
    """policy Condition{match{...}} -> condition tuple tree (compile/conditions.go:24-59)."""
    if cond is None:
        return None
    if "match" not in cond:
        raise CompileError("unsupported condition (only `match` is supported)")
    return _compile_match(cond["match"])


def _compile_match(m):
    if "expr" in m:
        text = m["expr"]
        celparser.parse(text)  # surface syntax errors at compile time
        return ("expr", text)
    for op in ("all", "any", "none"):
        if op in m:
            return (op, tuple(_compile_match(x) for x in (m[op].get("of") or [])))
    raise CompileError(f"unknown match operation: {sorted(m)}")


def condition_exprs(cond):
    """All expression texts in a condition tree."""
    if cond is None:
        return
    if cond[0] == "expr":
        yield cond[1]
    else:
        for c in cond[1]:
            yield from condition_exprs(c)


def _references(text: str):
    """(constants, variables) referenced as C.x / constants.x / V.x / variables.x
    (compile/variables.go:205-234)."""
    consts, vars_ = set(), set()
    for n in celparser.walk(celparser.parse(text)):
        if n[0] in ("select", "has") and n[1][0] == "ident":
            base = n[1][1]
            if base in ("C", "constants"):
                consts.add(n[2])
            elif base in ("V", "variables"):
                vars_.add(n[2])
    return consts, vars_


class _Defs:
    """Constants + variables visible to one policy module, with usage tracking."""

    def __init__(self, policies: dict, pol: dict):
        body = pol[policy_kind(pol)]
        self.consts = {}
        self.vars = {}  # name -> expr text
        cdef = body.get("constants") or {}
        for imp in cdef.get("import") or []:
            ec = policies.get(namer.export_constants_fqn(imp))
            if ec is None:
                raise CompileError(f"Constants import '{imp}' cannot be found")
            self.consts.update(ec["exportConstants"].get("definitions") or {})
        self.consts.update(cdef.get("local") or {})
        vdef = body.get("variables") or {}
        for imp in vdef.get("import") or []:
            ev = policies.get(namer.export_variables_fqn(imp))
            if ev is None:
                raise CompileError(f"Variables import '{imp}' cannot be found")
            self.vars.update(ev["exportVariables"].get("definitions") or {})
        self.vars.update(vdef.get("local") or {})
        self.vars.update(pol.get("variables") or {})  # deprecated top-level variables
        self.vars = {k: str(v) for k, v in self.vars.items()}
        self.deps = {}
        for name, text in self.vars.items():
            _, vs = _references(text)
            self.deps[name] = {v for v in vs if v in self.vars and v != name}
        self.reset()

    def reset(self):
        self.used_consts = set()
        self.used_vars = set()

    def use_expr(self, text: str):
        cs, vs = _references(text)
        for c in cs:
            if c in self.consts:
                self.used_consts.add(c)
        for v in vs:
            self._use_var(v)

    def use_condition(self, cond):
        for text in condition_exprs(cond):
            self.use_expr(text)

    def _use_var(self, name):
        if name in self.used_vars or name not in self.vars:
            return
        self.used_vars.add(name)
        cs, _ = _references(self.vars[name])
        for c in cs:
            if c in self.consts:
                self.used_consts.add(c)
        for d in self.deps[name]:
            self._use_var(d)

    def used_constants(self) -> dict:
        return {k: self.consts[k] for k in sorted(self.used_consts)}

    def ordered_variables(self):
        """Used variables in dependency order, ties broken by name
        (variables.go:281-305 - topo.SortStabilized by name)."""
        remaining = {n: set(self.deps[n]) for n in self.vars}
        order = []
        while remaining:
            ready = sorted(n for n, d in remaining.items() if not (d & remaining.keys()))
            if not ready:
                raise CompileError("variables form a cycle: %s" % sorted(remaining))
            n = ready[0]
            order.append(n)
            del remaining[n]
        return [(n, self.vars[n]) for n in order if n in self.used_vars]


def _compile_output(out, defs: _Defs):
    """compile.go:430-462"""
    if out is None:
        return None
    when = {}
    if out.get("expr"):
        when["rule_activated"] = out["expr"]
    w = out.get("when") or {}
    if w.get("ruleActivated"):
        when["rule_activated"] = w["ruleActivated"]
    if w.get("conditionNotMet"):
        when["condition_not_met"] = w["conditionNotMet"]
    for t in when.values():
        defs.use_expr(t)
    return when


def _effect(e):
    if e not in _EFFECTS:
        raise CompileError(f"invalid effect {e!r}")
    return _EFFECTS[e]


def _compile_derived_roles(policies: dict, name: str):
    """compile.go:329-367 -> {role name: runnable derived role}"""
    pol = policies.get(namer.derived_roles_fqn(name))
    if pol is None:
        raise CompileError(f"Derived roles import {name!r} cannot be found")
    defs = _Defs(policies, pol)
    out = {}
    for d in pol["derivedRoles"].get("definitions") or []:
        parents = []
        for pr in d.get("parentRoles") or []:
            if pr == ANY_ROLE:
                parents = [ANY_ROLE]
                break
            if pr not in parents:
                parents.append(pr)
        defs.reset()
        cond = compile_condition(d.get("condition"))
        defs.use_condition(cond)
        out[d["name"]] = {
            "name": d["name"],
            "parent_roles": parents,
            "origin_fqn": namer.derived_roles_fqn(pol["derivedRoles"]["name"]),
            "condition": cond,
            "constants": defs.used_constants(),
            "ordered_variables": defs.ordered_variables(),
        }
    return out


def compile_resource_policy(policies: dict, pol: dict) -> dict:
    """compile.go:197-245, 389-428 (only this policy: parent scopes are separate sets)."""
    rp = pol["resourcePolicy"]
    fqn = namer.resource_policy_fqn(rp["resource"], str(rp.get("version", "")), rp.get("scope", "") or "")
    role_imports = {}
    for imp in rp.get("importDerivedRoles") or []:
        for n, dr in _compile_derived_roles(policies, imp).items():
            role_imports.setdefault(n, []).append(dr)
    referenced = {}
    for rule in rp.get("rules") or []:
        for r in rule.get("derivedRoles") or []:
            imp = role_imports.get(r)
            if not imp:
                raise CompileError(f"Derived role {r!r} is not defined in any imports ({fqn})")
            if len(imp) > 1:
                raise CompileError(f"Derived role {r!r} is defined in more than one import ({fqn})")
            referenced[r] = imp[0]

    defs = _Defs(policies, pol)
    rules = []
    for i, rule in enumerate(rp.get("rules") or []):
        name = namer.resource_rule_name(rule.get("name", "") or "", i + 1)
        if not (rule.get("roles") or rule.get("derivedRoles")):
            raise CompileError(f"Rule '{name}' does not specify any roles or derived roles to be matched")
        cond = compile_condition(rule.get("condition"))
        defs.use_condition(cond)
        roles = []
        for r in rule.get("roles") or []:
            if r == ANY_ROLE:
                roles = [ANY_ROLE]
                break
            if r not in roles:
                roles.append(r)
        rules.append({
            "name": name,
            "actions": list(dict.fromkeys(rule.get("actions") or [])),
            "roles": roles,
            "derived_roles": list(dict.fromkeys(rule.get("derivedRoles") or [])),
            "condition": cond,
            "effect": _effect(rule.get("effect")),
            "emit_output": _compile_output(rule.get("output"), defs),
        })
    sp = _SP_NAMES[rp.get("scopePermissions")]
    return {
        "kind": "resource",
        "fqn": fqn,
        "resource": rp["resource"],
        "version": str(rp.get("version", "")),
        "scope": rp.get("scope", "") or "",
        "scope_permissions": sp,  # raw (possibly UNSPECIFIED); see ruletable rows
        "derived_roles": referenced,
        "rules": rules,
        "constants": defs.used_constants(),
        "ordered_variables": defs.ordered_variables(),
    }


def compile_principal_policy(policies: dict, pol: dict) -> dict:
    """compile.go:505-553"""
    pp = pol["principalPolicy"]
    fqn = namer.principal_policy_fqn(pp["principal"], str(pp.get("version", "")), pp.get("scope", "") or "")
    defs = _Defs(policies, pol)
    resource_rules = {}
    for rule in pp.get("rules") or []:
        action_rules = []
        for i, a in enumerate(rule.get("actions") or []):
            name = namer.principal_resource_action_rule_name(a.get("name", "") or "", rule["resource"], i + 1)
            cond = compile_condition(a.get("condition"))
            defs.use_condition(cond)
            action_rules.append({
                "action": a["action"],
                "name": name,
                "effect": _effect(a.get("effect")),
                "condition": cond,
                "emit_output": _compile_output(a.get("output"), defs),
            })
        # a later rule for the same resource replaces the earlier one (compile.go:545)
        resource_rules.pop(rule["resource"], None)
        resource_rules[rule["resource"]] = action_rules
    return {
        "kind": "principal",
        "fqn": fqn,
        "principal": pp["principal"],
        "version": str(pp.get("version", "")),
        "scope": pp.get("scope", "") or "",
        "scope_permissions": _SP_NAMES[pp.get("scopePermissions")],
        "resource_rules": resource_rules,
        "constants": defs.used_constants(),
        "ordered_variables": defs.ordered_variables(),
    }


def compile_role_policy(policies: dict, pol: dict) -> dict:
    """compile.go:77-137"""
    rp = pol["rolePolicy"]
    version = str(rp.get("version", "") or "") or namer.DEFAULT_VERSION
    scope = rp.get("scope", "") or ""
    fqn = namer.role_policy_fqn(rp["role"], version, scope)
    defs = _Defs(policies, pol)
    resources = {}
    for r in rp.get("rules") or []:
        cond = compile_condition(r.get("condition"))
        defs.use_condition(cond)
        resources.setdefault(r["resource"], []).append({
            "resource": r["resource"],
            "name": r.get("name", "") or "",
            "allow_actions": list(dict.fromkeys(r.get("allowActions") or [])),
            "condition": cond,
            "emit_output": _compile_output(r.get("output"), defs),
        })
    return {
        "kind": "role",
        "fqn": fqn,
        "role": rp["role"],
        "version": version,
        "scope": scope,
        "parent_roles": list(rp.get("parentRoles") or []),
        "resources": resources,
        "constants": defs.used_constants(),
        "ordered_variables": defs.ordered_variables(),
    }


def compile_all(policies: dict) -> list:
    """Every runnable policy set of a store, in FQN order (deterministic stand-in for the
    reference loader's unspecified order; effects do not depend on it)."""
    out = []
    for fqn in sorted(policies):
        pol = policies[fqn]
        k = policy_kind(pol)
        if k == "resourcePolicy":
            out.append(compile_resource_policy(policies, pol))
        elif k == "principalPolicy":
            out.append(compile_principal_policy(policies, pol))
        elif k == "rolePolicy":
            out.append(compile_role_policy(policies, pol))
    return out
