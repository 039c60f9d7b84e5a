"""Policy dicts -> runnable policy sets (the compile step that feeds the rule table).

Restates the reference's ``internal/compile``: a compilation unit is a policy with the definitions it needs (its scope
ancestors, the derived roles / constants / variables it imports); compiling it either gives the runnable policies of the unit or
the full SET of errors - kind, description, path and, when the YAML text is known (``policy/source.py``), line and column - that
the reference reports (``compile.go``, ``variables.go``, ``constants.go``, ``conditions.go``, ``errors.go``; file:line per
function below).  Pinned by the reference's own cases: tests/golden/compile_cases.json (tools/make_golden_compile.py),
tests/test_compile_cases.py.

Not restated: schema references are not loaded (``compile.go:358-376`` - schema validation is out of scope), and of cel-go's
type checker only two checks exist (``cel/check.py``), so some ill-typed expressions the reference rejects pass here.

A condition is a tuple tree:  ('expr', text) | ('all'|'any'|'none', (cond, ...)).
"""
from __future__ import annotations

import re
from collections import namedtuple

from .. import namer
from ..cel import check as celcheck
from ..cel import parser as celparser
from .loader import policy_fqn, policy_kind
from .source import Source, json_path

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

# errors.go:22-38, variables.go:27, constants.go:23
E_AMBIGUOUS_DERIVED_ROLE = "ambiguous derived role"
E_CONSTANT_REDEFINED = "constant redefined"
E_CYCLICAL_VARIABLES = "cyclical variable definitions"
E_EMPTY_OUTPUT = "empty output"
E_IMPORT_NOT_FOUND = "import not found"
E_INVALID_RESOURCE_RULE = "invalid resource rule"
E_MISSING_DEFINITION = "missing policy definition"
E_SCRIPTS_UNSUPPORTED = "scripts in conditions are no longer supported"
E_UNDEFINED_CONSTANT = "undefined constant"
E_UNDEFINED_VARIABLE = "undefined variable"
E_UNEXPECTED = "unexpected error"
E_UNKNOWN_DERIVED_ROLE = "unknown derived role"
E_VARIABLE_REDEFINED = "variable redefined"
E_INVALID_EXPRESSION = "invalid expression"
E_INVALID_VARIABLE_NAME = "invalid variable name"
E_INVALID_CONSTANT_NAME = "invalid constant name"

CompileErr = namedtuple("CompileErr", "file error description path line column")


class CompileError(ValueError):
    """errors.go:40-103 ErrorSet: every error of the unit(s), in the reference's order (file, line, column, description)."""

    def __init__(self, errors):
        self.errors = sorted(errors, key=lambda e: (e.file, e.line or 0, e.column or 0, e.description))
        lines = sorted(_error_string(e) for e in self.errors)
        ValueError.__init__(self, "%d compilation errors:\n%s" % (len(lines), "\n".join(lines)))


def _error_string(e):   # errors.go:151-159
    if e.description:
        if e.line:
            return "%s:%d:%d: %s (%s)" % (e.file, e.line, e.column, e.description, e.error)
        return "%s: %s (%s)" % (e.file, e.description, e.error)
    return "%s: %s" % (e.file, e.error)


def _goquote(s):
    """fmt's %q for the strings policies hold"""
    return '"' + s.replace("\\", "\\\\").replace('"', '\\"') + '"'


_IDENTIFIER = re.compile(r"^[_a-zA-Z][_a-zA-Z0-9]*\Z")
_KEYWORDS = ("false", "in", "null", "true")


def _identifier_error(name):
    """conditions/identifiers.go:24-34"""
    if name in _KEYWORDS:
        return "%s is a reserved keyword and can't be used as an identifier" % _goquote(name)
    if not _IDENTIFIER.match(name):
        return "%s is not a valid identifier" % _goquote(name)
    return None


# ---- conditions (conditions.go) ----------------------------------------------------------------------------------------------
def compile_condition(cond):
    """policy Condition{match{...}} -> condition tuple tree (conditions.go:24-59), without a module: the first problem raises."""
    unit = _Unit({}, None)
    mod = _Module(unit, {"apiVersion": "api.cerbos.dev/v1", "exportVariables": {"name": "UNKNOWN"}}, "UNKNOWN", None)
    out = mod.condition(("unknown",), cond, False)
    unit.raise_if_errors()
    return out


def condition_exprs(cond):
    """All expression texts in a condition tree."""
    if cond is None:
        return
    if cond[0] == "expr":
        yield cond[1]
    else:
        for c in cond[1]:
            yield from condition_exprs(c)


def _references(ast):
    """(constants, variables) selected as C.x / constants.x / V.x / variables.x (variables.go:205-234, constants.go:143-166)."""
    consts, vars_ = [], []
    if ast is None:
        return consts, vars_
    for n in celparser.walk(ast):
        if n[0] in ("select", "has") and n[1][0] == "ident":
            base = n[1][1]
            if base in ("C", "constants") and n[2] not in consts:
                consts.append(n[2])
            elif base in ("V", "variables") and n[2] not in vars_:
                vars_.append(n[2])
    return consts, vars_


# ---- the unit and its modules (context.go) -------------------------------------------------------------------------------------
class _Unit:
    def __init__(self, policies: dict, sources: dict | None):
        self.policies, self.sources = policies, sources or {}
        self.errors = {}   # errors.go:105-121: keyed by file, position and description - the same report twice is one error

    def module(self, fqn):
        pol = self.policies.get(fqn)
        if pol is None:
            return None
        return _Module(self, pol, fqn, self.sources.get(fqn))

    def add(self, err: CompileErr):
        self.errors[(err.file, err.line or 0, err.column or 0, err.description)] = err

    def raise_if_errors(self):
        if self.errors:
            raise CompileError(self.errors.values())


class _Module:
    """moduleCtx (context.go:47-85): one policy definition inside a unit."""

    def __init__(self, unit, pol, fqn, source):
        self.unit, self.pol, self.fqn = unit, pol, fqn
        self.kind = policy_kind(pol)
        self.body = pol[self.kind] or {}
        self.source = source if source is not None else Source(_source_file(pol, fqn))
        self.file = self.source.file
        self.constants = self.variables = None

    # -- reporting
    def err(self, kind, description):
        self.unit.add(CompileErr(self.file, kind, description, None, None, None))

    def err_at(self, path, at, kind, description):
        pos = self.source.position(path, at)
        self.unit.add(CompileErr(self.file, kind, description, json_path(path), pos[0] if pos else None, pos[1] if pos else None))

    def place(self, path, at="value"):
        return self.source.position(path, at)

    # -- expressions (conditions.go:61-86)
    def cel(self, path, at, text, mark_used):
        text = "" if text is None else str(text)
        ast, msgs = celcheck.compile_issues(text)
        if msgs:
            self.err_at(path, at, E_INVALID_EXPRESSION, "Invalid expression `%s`: [%s]" % (text, ", ".join(msgs)))
            return None
        if mark_used:
            self.constants.use(path, at, ast)
            self.variables.use(path, at, ast)
        return ast

    def condition(self, path, cond, mark_used):
        """conditions.go:24-36"""
        if cond is None:
            return None
        if "match" not in cond:
            self.err_at(path, "key", E_SCRIPTS_UNSUPPORTED, "Unsupported feature")
            return None
        return self.match(path + ("match",), cond["match"], mark_used)

    def match(self, path, m, mark_used):
        """conditions.go:38-59"""
        if m is None:
            return None
        if "expr" in m:
            text = "" if m["expr"] is None else str(m["expr"])
            self.cel(path + ("expr",), "key", text, mark_used)
            return ("expr", text)
        for op in ("all", "any", "none"):
            if op in m:
                of = (m[op] or {}).get("of") or []
                return (op, tuple(self.match(path + (op, "of", i), x, mark_used) for i, x in enumerate(of)))
        self.err_at(path, "key", E_UNEXPECTED, "Unknown match operation: %s" % sorted(m))
        return None

    def output(self, path, out):
        """compile.go:430-462"""
        if out is None:
            return None
        when = {}
        if out.get("expr"):
            self.cel(path + ("output", "expr"), "key", out["expr"], True)
            when["rule_activated"] = out["expr"]
        w = out.get("when") or {}
        if w.get("ruleActivated"):
            self.cel(path + ("output", "when", "ruleActivated"), "key", w["ruleActivated"], True)
            when["rule_activated"] = w["ruleActivated"]
        if w.get("conditionNotMet"):
            self.cel(path + ("output", "when", "conditionNotMet"), "key", w["conditionNotMet"], True)
            when["condition_not_met"] = w["conditionNotMet"]
        if not when:
            self.err_at(path + ("output",), "key", E_EMPTY_OUTPUT, "output must have at least one expression")
        return when

    # -- constants and variables of the module
    def compile_definitions(self):
        """compilePolicyConstants (constants.go:25-51) then compilePolicyVariables (variables.go:29-58)"""
        if self.constants is not None:
            return
        cdef, vdef = self.body.get("constants") or {}, self.body.get("variables") or {}
        self.constants = _Constants(self)
        for i, imp in enumerate(cdef.get("import") or []):
            ec = self.unit.module(namer.export_constants_fqn(imp))
            if ec is None:
                self.err_at((self.kind, "constants", "import", i), "value", E_IMPORT_NOT_FOUND, "Constants import '%s' cannot be found" % imp)
                continue
            ec.compile_exported()
            self.constants.import_from(ec, "import '%s'" % imp)
        self.constants.compile(cdef.get("local"), (self.kind, "constants", "local"), "policy local constants")
        self.constants.report_redefined()

        self.variables = _Variables(self)
        for i, imp in enumerate(vdef.get("import") or []):
            ev = self.unit.module(namer.export_variables_fqn(imp))
            if ev is None:
                self.err_at((self.kind, "variables", "import", i), "value", E_IMPORT_NOT_FOUND, "Variables import '%s' cannot be found" % imp)
                continue
            ev.compile_exported()
            self.variables.import_from(ev, "import '%s'" % imp)
        self.variables.compile(vdef.get("local"), (self.kind, "variables", "local"), "policy local variables")
        self.variables.compile(self.pol.get("variables"), ("variables",), "deprecated top-level policy variables")
        self.variables.resolve()

    def compile_exported(self):
        """compileExportConstants (constants.go:53-68) / compileExportVariables (variables.go:60-75)"""
        if self.kind == "exportConstants" and self.constants is None:
            self.constants = _Constants(self)
            self.constants.compile(self.body.get("definitions"), (self.kind, "definitions"), "definitions")
        elif self.kind == "exportVariables" and self.variables is None:
            self.constants = self.constants or _Constants(self)
            self.variables = _Variables(self)
            self.variables.compile(self.body.get("definitions"), (self.kind, "definitions"), "definitions")


def _source_file(pol, fqn):
    meta = pol.get("metadata") or {}
    return meta.get("sourceFile") or "%s.yaml" % namer.policy_key_from_fqn(fqn)


_Def = namedtuple("_Def", "name value ast module path source")


def _places(defs):
    """variableDefinitionPlaces / constantDefinitionPlaces (variables.go:160-172, constants.go:127-139)"""
    out = []
    for d in defs:
        pos = d.module.place(d.path)
        out.append("%s (%s:%d:%d)" % (d.source, d.module.file, pos[0], pos[1]) if pos else "%s (%s)" % (d.source, d.module.file))
    return " and ".join(out) if len(out) == 2 else "%s, and %s" % (", ".join(out[:-1]), out[-1])


class _Constants:
    """constantDefinitions (constants.go:70-196)"""

    def __init__(self, mod):
        self.mod, self.values, self.sources, self.used = mod, {}, {}, set()

    def compile(self, definitions, path, source):
        for name, value in (definitions or {}).items():
            name = _key_text(name)
            p = path + (name,)
            bad = _identifier_error(name)
            if bad:
                self.mod.err_at(p, "key", E_INVALID_CONSTANT_NAME, bad)
            self.add(_Def(name, value, None, self.mod, p, source))

    def import_from(self, other, source):
        for name in other.constants.values:
            self.add(other.constants.sources[name][0]._replace(source=source))

    def add(self, d):
        self.values[d.name] = d.value
        self.sources.setdefault(d.name, []).append(d)

    def report_redefined(self):
        for name, defs in self.sources.items():
            if len(defs) > 1:
                self.mod.err(E_CONSTANT_REDEFINED, "Constant '%s' has multiple definitions in %s" % (name, _places(defs)))

    def use(self, path, at, ast):
        for name in _references(ast)[0]:
            if name in self.values:
                self.used.add(name)
            else:
                self.mod.err_at(path, at, E_UNDEFINED_CONSTANT, "Undefined constant '%s'" % name)

    def reset_usage(self):
        self.used = set()

    def used_values(self) -> dict:
        return {k: self.values[k] for k in sorted(self.used)}


class _Variables:
    """variableDefinitions (variables.go:96-359)"""

    def __init__(self, mod):
        self.mod, self.defs, self.latest, self.sources, self.used = mod, [], {}, {}, set()
        self.deps = {}   # index of a definition -> indices of the definitions it reads

    def compile(self, definitions, path, source):
        for name, text in (definitions or {}).items():
            name = _key_text(name)
            p = path + (name,)
            bad = _identifier_error(name)
            if bad:
                self.mod.err_at(p, "key", E_INVALID_VARIABLE_NAME, bad)
            text = "" if text is None else str(text)
            self.add(_Def(name, text, self.mod.cel(p, "value", text, False), self.mod, p, source))

    def import_from(self, other, source):
        for d in other.variables.defs:
            self.add(d._replace(source=source))

    def add(self, d):
        self.sources.setdefault(d.name, []).append(d)
        self.latest[d.name] = len(self.defs)
        self.defs.append(d)

    def resolve(self):
        """variables.go:133-203"""
        for name, defs in self.sources.items():
            if len(defs) > 1:
                self.mod.err(E_VARIABLE_REDEFINED, "Variable '%s' has multiple definitions in %s" % (name, _places(defs)))
        for name, i in self.latest.items():
            d = self.defs[i]
            consts, vars_ = _references(d.ast)
            for c in consts:
                if c not in self.mod.constants.values:
                    d.module.err_at(d.path, "value", E_UNDEFINED_CONSTANT, "Undefined constant '%s' referenced in variable '%s'" % (c, name))
            for v in vars_:
                if v == name:
                    d.module.err_at(d.path, "value", E_CYCLICAL_VARIABLES, "Variable '%s' references itself" % v)
                elif v not in self.latest:
                    d.module.err_at(d.path, "value", E_UNDEFINED_VARIABLE, "Undefined variable '%s' referenced in variable '%s'" % (v, name))
                else:
                    self.deps.setdefault(i, set()).add(self.latest[v])
        self.reset_usage()

    def reset_usage(self):
        self.used = set()

    def use(self, path, at, ast):
        """variables.go:240-267"""
        for name in _references(ast)[1]:
            if name in self.latest:
                self._use(self.latest[name])
            else:
                self.mod.err_at(path, at, E_UNDEFINED_VARIABLE, "Undefined variable '%s'" % name)

    def _use(self, i):
        d = self.defs[i]
        if d.name in self.used:
            return
        self.used.add(d.name)
        self.mod.constants.use(d.path, "value", d.ast)
        for j in self.deps.get(i, ()):
            self._use(j)

    def ordered(self):
        """Used variables in dependency order, ties by name (variables.go:269-305: topo.SortStabilized); the strongly connected
        groups are reported as cycles (variables.go:307-353) and nothing is returned."""
        n = len(self.defs)
        groups = _strongly_connected(n, self.deps)
        cycles = [g for g in groups if len(g) > 1]
        if cycles:
            for g in cycles:
                members = sorted((self.defs[i] for i in g), key=lambda d: d.name)
                parts = []
                for d in members:
                    pos = d.module.place(d.path)
                    parts.append("'%s' (%s:%d:%d)" % (d.name, d.module.file, pos[0], pos[1]) if pos else "'%s'" % d.name)
                text = " and ".join(parts) if len(parts) == 2 else "%s, and %s" % (", ".join(parts[:-1]), parts[-1])
                members[0].module.err_at(members[0].path, "value", E_CYCLICAL_VARIABLES, "Variables %s form a cycle" % text)
            return []
        remaining = {i: set(self.deps.get(i, ())) for i in range(n)}
        order = []
        while remaining:
            i = min((i for i, d in remaining.items() if not (d & remaining.keys())), key=lambda i: (self.defs[i].name, i))
            order.append(i)
            del remaining[i]
        return [(self.defs[i].name, self.defs[i].value) for i in order if self.defs[i].name in self.used]


def _strongly_connected(n, deps):
    """Tarjan's algorithm, iteratively: the groups of definitions that reach each other."""
    index, low, on, stack, out, counter = {}, {}, set(), [], [], [0]
    for root in range(n):
        if root in index:
            continue
        work = [(root, iter(sorted(deps.get(root, ()))))]
        index[root] = low[root] = counter[0]
        counter[0] += 1
        stack.append(root)
        on.add(root)
        while work:
            v, it = work[-1]
            advanced = False
            for w in it:
                if w not in index:
                    index[w] = low[w] = counter[0]
                    counter[0] += 1
                    stack.append(w)
                    on.add(w)
                    work.append((w, iter(sorted(deps.get(w, ())))))
                    advanced = True
                    break
                if w in on:
                    low[v] = min(low[v], index[w])
            if advanced:
                continue
            work.pop()
            if work:
                low[work[-1][0]] = min(low[work[-1][0]], low[v])
            if low[v] == index[v]:
                group = []
                while True:
                    w = stack.pop()
                    on.discard(w)
                    group.append(w)
                    if w == v:
                        break
                out.append(group)
    return out


def _key_text(k):
    """A YAML key the loader resolved to a non-string (`true:`, `1:`) as the text the reference's map<string, ...> holds."""
    if isinstance(k, str):
        return k
    if k is None:
        return "null"
    if isinstance(k, bool):
        return "true" if k else "false"
    return str(k)


def _unique(items, any_collapses=False):
    out = []
    for x in items or []:
        if any_collapses and x == ANY_ROLE:
            return [ANY_ROLE]
        if x not in out:
            out.append(x)
    return out


def _effect(mod, e):
    if e not in _EFFECTS:
        mod.err(E_UNEXPECTED, "invalid effect %r" % (e,))
        return "DENY"
    return _EFFECTS[e]


# ---- derived roles (compile.go:329-367) ------------------------------------------------------------------------------------------
def _compile_derived_roles(mod: _Module) -> dict:
    mod.compile_definitions()
    dr = mod.body
    out = {}
    for i, d in enumerate(dr.get("definitions") or []):
        mod.constants.reset_usage()
        mod.variables.reset_usage()
        cond = mod.condition(("derivedRoles", "definitions", i, "condition"), d.get("condition"), True)
        out[d["name"]] = {
            "name": d["name"],
            "parent_roles": _unique(d.get("parentRoles"), any_collapses=True),
            "origin_fqn": namer.derived_roles_fqn(dr["name"]),
            "condition": cond,
            "constants": mod.constants.used_values(),
            "ordered_variables": mod.variables.ordered(),
        }
    return out


def _imported_derived_roles(mod: _Module):
    """compile.go:252-327 -> {role: runnable derived role} for the roles the rules name, or None when the unit has errors."""
    rp = mod.body
    role_imports = {}
    for i, imp in enumerate(rp.get("importDerivedRoles") or []):
        path = ("resourcePolicy", "importDerivedRoles", i)
        dr_mod = mod.unit.module(namer.derived_roles_fqn(imp))
        if dr_mod is None or dr_mod.kind != "derivedRoles":
            mod.err_at(path, "value", E_IMPORT_NOT_FOUND, "Derived roles import %s cannot be found" % _goquote(imp))
            continue
        for name, compiled in _compile_derived_roles(dr_mod).items():
            role_imports.setdefault(name, []).append((imp, dr_mod.file, compiled, path))
    referenced, unknown, ambiguous = {}, {}, {}
    for i, rule in enumerate(rp.get("rules") or []):
        for j, r in enumerate(rule.get("derivedRoles") or []):
            imps = role_imports.get(r)
            if not imps:
                unknown[r] = ("resourcePolicy", "rules", i, "derivedRoles", j)
            elif len(imps) > 1:
                if r not in ambiguous:
                    places = []
                    for imp, file, _, path in imps:
                        pos = mod.place(path)
                        places.append("%s (imported as %s at %d:%d)" % (file, _goquote(imp), pos[0], pos[1]) if pos
                                      else "%s (imported as %s)" % (file, _goquote(imp)))
                    ambiguous[r] = ", ".join(places)
            else:
                referenced[r] = imps[0][2]
    for r, path in unknown.items():
        mod.err_at(path, "value", E_UNKNOWN_DERIVED_ROLE, "Derived role %s is not defined in any imports" % _goquote(r))
    for r, places in ambiguous.items():
        mod.err(E_AMBIGUOUS_DERIVED_ROLE, "Derived role %s is defined in more than one import: %s" % (_goquote(r), places))
    return None if mod.unit.errors else referenced


# ---- resource policies (compile.go:139-250, 378-428) ---------------------------------------------------------------------------
def _compile_resource_policy(mod: _Module):
    rp = mod.body
    referenced = _imported_derived_roles(mod)
    if referenced is None:
        return None
    mod.compile_definitions()
    rules = []
    for i, rule in enumerate(rp.get("rules") or []):
        path = ("resourcePolicy", "rules", i)
        name = namer.resource_rule_name(rule.get("name", "") or "", i + 1)
        if not (rule.get("roles") or rule.get("derivedRoles")):
            mod.err_at(path, "colon", E_INVALID_RESOURCE_RULE, "Rule '%s' does not specify any roles or derived roles to be matched" % name)
        rules.append({
            "name": name,
            "actions": _unique(rule.get("actions")),
            "roles": _unique(rule.get("roles"), any_collapses=True),
            "derived_roles": _unique(rule.get("derivedRoles")),
            "condition": mod.condition(path + ("condition",), rule.get("condition"), True),
            "effect": _effect(mod, rule.get("effect")),
            "emit_output": mod.output(path, rule.get("output")),
        })
    version, scope = str(rp.get("version", "")), rp.get("scope", "") or ""
    return {
        "kind": "resource",
        "fqn": namer.resource_policy_fqn(rp["resource"], version, scope),
        "resource": rp["resource"],
        "version": version,
        "scope": scope,
        "scope_permissions": _SP_NAMES[rp.get("scopePermissions")],  # raw (possibly UNSPECIFIED); see ruletable rows
        "derived_roles": referenced,
        "rules": rules,
        "constants": mod.constants.used_values(),
        "ordered_variables": mod.variables.ordered(),
    }


# ---- principal policies (compile.go:464-553) -----------------------------------------------------------------------------------
def _compile_principal_policy(mod: _Module):
    pp = mod.body
    mod.compile_definitions()
    resource_rules = {}
    for n, rule in enumerate(pp.get("rules") or []):
        action_rules = []
        for i, a in enumerate(rule.get("actions") or []):
            path = ("principalPolicy", "rules", n, "actions", i)
            action_rules.append({
                "action": a["action"],
                "name": namer.principal_resource_action_rule_name(a.get("name", "") or "", rule["resource"], i + 1),
                "effect": _effect(mod, a.get("effect")),
                "condition": mod.condition(path + ("condition",), a.get("condition"), True),
                "emit_output": mod.output(path, a.get("output")),
            })
        # a later rule for the same resource replaces the earlier one (compile.go:545)
        resource_rules.pop(rule["resource"], None)
        resource_rules[rule["resource"]] = action_rules
    version, scope = str(pp.get("version", "")), pp.get("scope", "") or ""
    return {
        "kind": "principal",
        "fqn": namer.principal_policy_fqn(pp["principal"], version, scope),
        "principal": pp["principal"],
        "version": version,
        "scope": scope,
        "scope_permissions": _SP_NAMES[pp.get("scopePermissions")],
        "resource_rules": resource_rules,
        "constants": mod.constants.used_values(),
        "ordered_variables": mod.variables.ordered(),
    }


# ---- role policies (compile.go:77-137) ---------------------------------------------------------------------------------------------
def _compile_role_policy(mod: _Module):
    rp = mod.body
    mod.compile_definitions()
    version = str(rp.get("version", "") or "") or namer.DEFAULT_VERSION
    scope = rp.get("scope", "") or ""
    resources = {}
    for i, r in enumerate(rp.get("rules") or []):
        path = ("rolePolicy", "rules", i)
        resources.setdefault(r["resource"], []).append({
            "resource": r["resource"],
            "name": r.get("name", "") or "",
            "allow_actions": _unique(r.get("allowActions")),
            "condition": mod.condition(path + ("condition",), r.get("condition"), True),
            "emit_output": mod.output(path, r.get("output")),
        })
    return {
        "kind": "role",
        "fqn": namer.role_policy_fqn(rp["role"], version, scope),
        "role": rp["role"],
        "version": version,
        "scope": scope,
        "parent_roles": list(rp.get("parentRoles") or []),
        "resources": resources,
        "constants": mod.constants.used_values(),
        "ordered_variables": mod.variables.ordered(),
    }


_COMPILERS = {"resourcePolicy": _compile_resource_policy, "principalPolicy": _compile_principal_policy, "rolePolicy": _compile_role_policy}


def _ancestor_fqns(pol):
    """namer_non_embedded.go:72-98 FQNTree without the policy itself: the same policy at every parent scope, then without a scope."""
    kind = policy_kind(pol)
    body = pol[kind]
    scope = body.get("scope", "") or ""
    if kind == "resourcePolicy":
        base = namer.resource_policy_fqn(body["resource"], str(body.get("version", "")), "")
    elif kind == "principalPolicy":
        base = namer.principal_policy_fqn(body["principal"], str(body.get("version", "")), "")
    else:
        return []   # role policies don't functionally have ancestors
    return [namer.with_scope(base, s) for s in namer.scope_parents(scope)]


def _compile_unit(unit: _Unit, fqn: str, require_ancestors: bool = True):
    """Compile (compile.go:52-75): the runnable policies of the unit - the policy, then its scope ancestors - or None."""
    mod = unit.module(fqn)
    if mod is None:
        raise KeyError(fqn)
    compiler = _COMPILERS.get(mod.kind)
    if compiler is None:
        return []   # derived roles and exports compile with the policies that import them
    first = compiler(mod)
    if first is None:
        return None
    out = [first]
    ancestors = _ancestor_fqns(mod.pol)
    if not require_ancestors:
        ancestors = [a for a in ancestors if a in unit.policies]
    for a in ancestors:
        anc = unit.module(a)
        if anc is None:   # reportMissingAncestors (compile.go:555-564): every missing one, once the walk meets the first
            for m in ancestors:
                if m not in unit.policies:
                    mod.err(E_MISSING_DEFINITION, "Missing ancestor policy %s" % _goquote(namer.policy_key_from_fqn(m)))
            return None
        compiled = compiler(anc)
        if compiled is None:
            return None
        out.append(compiled)
    return out


def compile_unit(policies: dict, fqn: str, sources: dict | None = None) -> list:
    """One compilation unit: [the policy, its scope ancestors from the nearest to the root] as runnable dicts; every error of the
    unit in one CompileError.  `sources`: {fqn: policy.source.Source} for positions in the errors."""
    unit = _Unit(policies, sources)
    out = _compile_unit(unit, fqn)
    unit.raise_if_errors()
    return out


def compile_resource_policy(policies: dict, pol: dict) -> dict:
    return compile_unit(policies, policy_fqn(pol))[0]


def compile_principal_policy(policies: dict, pol: dict) -> dict:
    return compile_unit(policies, policy_fqn(pol))[0]


def compile_role_policy(policies: dict, pol: dict) -> dict:
    return compile_unit(policies, policy_fqn(pol))[0]


def compile_all(policies: dict, sources: dict | None = None, require_ancestors: bool = True) -> list:
    """Every runnable policy of a store, in FQN order (deterministic stand-in for the reference loader's unspecified order;
    effects do not depend on it); the errors of all units together (BatchCompile, compile.go:39-49).
    `require_ancestors=False` lets a scoped policy stand without the same policy at its parent scopes - the reference never does
    (compile.go:555-564); generated test stores leave scopes out to put holes into the scope chains the kernels walk."""
    out, errors = [], {}
    for fqn in sorted(policies):
        if policy_kind(policies[fqn]) not in _COMPILERS:
            continue
        unit = _Unit(policies, sources)
        compiled = _compile_unit(unit, fqn, require_ancestors)
        errors.update(unit.errors)
        if compiled:
            out.append(compiled[0])
    if errors:
        raise CompileError(errors.values())
    return out
