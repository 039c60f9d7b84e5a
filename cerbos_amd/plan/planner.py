"""PlanResources: for a principal, actions and a resource KIND, the condition under which a resource of that kind is allowed - ALWAYS_ALLOWED,
ALWAYS_DENIED or a filter over the resource's (unknown) attributes.  The walk of internal/ruletable/plan.go:31-415 over the rule table:
per action, resource policies then principal policies, per role, per scope - every binding's condition partially evaluated
(cerbos_amd/plan/partial.py) into a node, the nodes combined as the scopes' permissions, role policies and effects say - and the
filter of cerbos_amd/plan/filter.py made of the result.  Host-side and symbolic, like the reference's: the query planner is not the
batched hot path."""
from __future__ import annotations

import json

from .. import namer
from ..cel.parser import parse
from ..policy.compile import SP_OVERRIDE_PARENT, SP_REQUIRE_PARENTAL_CONSENT
from ..ruletable.build import KIND_PRINCIPAL, KIND_RESOURCE
from . import filter as flt
from .index import PlanIndex
from .partial import CelEvalError, Partial, inline_variables, request_env

TRUE, FALSE = ("expr", ("lit", "bool", True)), ("expr", ("lit", "bool", False))


class StrictEvaluationError(Exception):
    pass


def const_bool(node):
    """IsNodeConstBool (planner.go:121-136) -> (is a constant, its value)"""
    if node is not None and node[0] == "expr" and node[1][0] == "lit" and node[1][1] == "bool":
        return True, bool(node[1][2])
    return False, False


def _hash(node):
    return json.dumps(node, sort_keys=True, default=str)


def _dedup(nodes):
    seen, out = set(), []
    for n in nodes:
        h = _hash(n)
        if h not in seen:
            seen.add(h)
            out.append(n)
    return out


def mk_or(nodes):
    u = _dedup(nodes)
    return None if not u else u[0] if len(u) == 1 else ("or", u)


def mk_and(nodes):
    u = _dedup(nodes)
    return None if not u else u[0] if len(u) == 1 else ("and", u)


def invert(node):
    """InvertNodeBooleanValue (planner.go:283-304)"""
    if node[0] == "not":
        return node[1][0] if len(node[1]) == 1 else ("and", list(node[1]))
    return ("not", [node])


def add_node(cur, nxt, combine):
    if nxt is None:
        return cur
    if cur is None:
        return nxt
    return combine([cur, nxt])


def gate(child_override_allow, deny):
    """gateByChildOverrideAllow (plan.go:418-434)"""
    if deny is None or child_override_allow is None:
        return deny
    inv = invert(child_override_allow)
    isc, bv = const_bool(deny)
    if isc and bv:
        return inv
    return mk_and([inv, deny])


class Planner:
    def __init__(self, rt: dict):
        self.rt = rt
        self.idx = PlanIndex(rt)
        self.principal_scopes = set(rt["principal_scopes"])
        self.resource_scopes = set(rt["resource_scopes"])
        self._parsed = {}
        self._sources = {}

    # ruletable.go:848-882 GetAllScopes: the scopes of the chain that hold SOME policy of the kind (not this name's: the index answers that)
    def all_scopes(self, kind, scope, name, version, lenient):
        have = self.principal_scopes if kind == KIND_PRINCIPAL else self.resource_scopes
        make = namer.principal_policy_fqn if kind == KIND_PRINCIPAL else namer.resource_policy_fqn
        scopes, first = [], ""
        if scope in have:
            scopes.append(scope)
            first = make(name, version, scope)
        elif not lenient:
            return [], ""
        for s in namer.scope_parents(scope):
            if s in have:
                scopes.append(s)
                if not first:
                    first = make(name, version, s)
        return scopes, first

    def source_keys(self, fqn):
        """The keys of the policy set's SourceAttributes (compile.go:153-180, 474-494): the policy and, scoped, its ancestors"""
        keys = self._sources.get(fqn)
        if keys is None:
            meta = self.rt["meta"].get(fqn)
            keys = []
            if meta is not None:
                keys.append(namer.policy_key_from_fqn(fqn))
                if meta["kind"] in ("resource", "principal") and "/" in fqn:
                    make = namer.resource_policy_fqn if meta["kind"] == "resource" else namer.principal_policy_fqn
                    for anc in namer.scope_parents(fqn.split("/", 1)[1]):
                        a = make(meta["name"], meta["version"], anc)
                        if a in self.rt["meta"]:
                            keys.append(namer.policy_key_from_fqn(a))
            self._sources[fqn] = keys
        return keys

    def parse(self, text):
        t = self._parsed.get(text)
        if t is None:
            t = self._parsed[text] = parse(text)
        return t

    def variable_exprs(self, ordered):
        """VariableExprs: every variable's expression with the ones before it inlined"""
        out = {}
        for name, text in ordered or ():
            out[name] = inline_variables(self.parse(text), out)
        return out

    def plan(self, inp: dict, globals_=None, default_policy_version="default", default_scope="", lenient_scope_search=False,
             strict_evaluation=False, now_ns=None):   # noqa: C901
        principal, resource = inp["principal"], inp["resource"]
        actions = list(inp.get("actions") or ([inp["action"]] if inp.get("action") else []))
        p_scope = namer.scope_value(principal.get("scope", "") or default_scope)
        p_ver = principal.get("policyVersion", "") or default_policy_version
        r_scope = namer.scope_value(resource.get("scope", "") or default_scope)
        r_ver = resource.get("policyVersion", "") or default_policy_version
        out = {"requestId": inp.get("requestId", ""), "kind": resource.get("kind", ""), "policyVersion": resource.get("policyVersion", ""),
               "action": inp.get("action") or "", "actions": list(inp.get("actions") or []),   # as they came (planner.go:113-125)
               "scope": namer.scope_value(resource.get("scope", "")), "matchedScopes": {}, "evaluationErrors": []}
        p_scopes, _ = self.all_scopes(KIND_PRINCIPAL, p_scope, principal.get("id", ""), p_ver, lenient_scope_search)
        r_scopes, _ = self.all_scopes(KIND_RESOURCE, r_scope, resource.get("kind", ""), r_ver, lenient_scope_search)
        if not p_scopes and not r_scopes:
            out["filter"] = {"kind": "KIND_ALWAYS_DENIED"}
            out["filterDebug"] = "NO_MATCH"
            out["effectivePolicies"] = []
            return out
        touched = set()
        errors = []
        ev = _Cond(self, principal, resource, inp.get("auxData"), globals_, strict_evaluation, errors, now_ns)
        sres = namer.sanitize(resource.get("kind", ""))
        all_roles = set(self.idx.add_parent_roles([r_scope], principal.get("roles") or []))
        filters, policy_match = [], False
        for action in actions:
            out["matchedScopes"][action] = ""
            allow_nodes, deny_nodes = [], []
            dr_lists = {}
            has_pt_allow, root, eval_err = False, None, None
            try:
                for pt in (KIND_RESOURCE, KIND_PRINCIPAL):
                    pt_allow = pt_deny = None
                    scopes = p_scopes if pt == KIND_PRINCIPAL else r_scopes
                    for i, role in enumerate(principal.get("roles") or []):
                        if i > 0 and pt == KIND_PRINCIPAL:
                            break
                        role_allow = role_deny = role_deny_rp = None
                        pending_allow = False
                        child_override = None
                        roles_inc = self.idx.add_parent_roles([r_scope], [role])
                        for scope in scopes:
                            isc, bv = const_bool(child_override)
                            if isc and bv:
                                break
                            s_allow = s_deny = s_deny_rp = None
                            dr_list = None
                            if pt == KIND_RESOURCE:
                                if scope not in dr_lists:
                                    dr_lists[scope] = ev.derived_roles_list(resource.get("kind", ""), r_ver, scope, all_roles)
                                dr_list = dr_lists[scope]
                            pid = principal.get("id", "") if pt == KIND_PRINCIPAL else ""
                            for b in self.idx.query(r_ver, sres, scope, action, roles_inc, pt, pid):
                                touched.update(self.source_keys(b["origin_fqn"]))
                                consts, variables = {}, {}
                                if b.get("params") is not None:
                                    consts = b["params"]["constants"]
                                    variables = self.variable_exprs(b["params"]["ordered_variables"])
                                node = ev.condition(b["condition"], consts, variables, dr_list)
                                if b.get("derived_role_condition") is not None:
                                    dp = b.get("derived_role_params") or {"constants": {}, "ordered_variables": []}
                                    dr_node = ev.condition(b["derived_role_condition"], dp["constants"], self.variable_exprs(dp["ordered_variables"]), dr_list)
                                    node = dr_node if b["condition"] is None else ("and", [node, dr_node])
                                if b["effect"] == "ALLOW":
                                    s_allow = add_node(s_allow, node, mk_or)
                                elif b["effect"] == "DENY":
                                    isc, bv = const_bool(node)
                                    if isc and not bv:
                                        continue
                                    if b.get("from_role_policy"):
                                        s_deny_rp = add_node(s_deny_rp, node, mk_or)
                                    else:
                                        s_deny = add_node(s_deny, node, mk_or)
                            s_deny = gate(child_override, s_deny)
                            s_deny_rp = gate(child_override, s_deny_rp)
                            role_deny = add_node(role_deny, s_deny, mk_or)
                            role_deny_rp = add_node(role_deny_rp, s_deny_rp, mk_or)
                            sp = self.rt["scope_permissions"].get(scope, 0)
                            if s_allow is not None:
                                if role_allow is None:
                                    role_allow = s_allow
                                elif pending_allow:
                                    role_allow = mk_and([role_allow, s_allow])
                                    pending_allow = False
                                else:
                                    role_allow = mk_or([role_allow, s_allow])
                                if sp == SP_REQUIRE_PARENTAL_CONSENT:
                                    pending_allow = True
                            if (s_deny is not None or s_deny_rp is not None or s_allow is not None) and sp == SP_OVERRIDE_PARENT:
                                out["matchedScopes"][action] = scope
                            if s_allow is not None and sp == SP_OVERRIDE_PARENT:
                                child_override = add_node(child_override, s_allow, mk_or)
                        if pending_allow:
                            role_allow = None
                        const_true = False
                        for d in (role_deny, role_deny_rp):
                            isc, bv = const_bool(d)
                            if isc and bv:
                                const_true = True
                                break
                        if const_true:
                            role_allow, role_deny, role_deny_rp = FALSE, None, None
                        elif role_allow is not None and role_deny is None and role_deny_rp is None:
                            isc, bv = const_bool(role_allow)
                            if isc and bv:
                                pt_allow, pt_deny = role_allow, None
                                break
                        if role_deny_rp is not None and role_allow is not None:
                            role_allow = mk_and([role_allow, invert(role_deny_rp)])
                        pt_allow = add_node(pt_allow, role_allow, mk_or)
                        pt_deny = add_node(pt_deny, role_deny, mk_or)
                    if pt_allow is not None:
                        has_pt_allow = True
                        root = pt_allow if root is None else mk_or([pt_allow, root])
                    if pt_deny is not None:
                        inv = invert(pt_deny)
                        root = inv if root is None else mk_and([inv, root])
            except StrictEvaluationError as e:
                eval_err = e
            if eval_err is not None:
                policy_match = True
                allow_nodes, deny_nodes = [], [FALSE]
            elif root is not None:
                policy_match = True
                if not has_pt_allow:
                    deny_nodes = [FALSE]
                else:
                    allow_nodes.append(root)
            if not allow_nodes and deny_nodes:
                deny_nodes = [FALSE]
            filters.append(flt.to_filter(_node_filter_ast(allow_nodes, deny_nodes)))
        out["filter"] = flt.merge_with_and(filters)
        out["filterDebug"] = flt.filter_to_string(out["filter"]) if policy_match else "NO_MATCH"
        # CELErrors.All() (evaluator/cel_errors.go:84-118): an (expression, message) pair once, ordered by expression then message;
        # enginev1.EvaluationError in its protojson form, as a CheckOutput's
        out["evaluationErrors"] = [{"celError": {"expression": e, "message": m}} for e, m in sorted({(x["expr"], x["message"]) for x in errors})]
        out["effectivePolicies"] = sorted(touched)     # the call's AuditTrail.EffectivePolicies (plan.go:50-51, 201-203)
        return out


def _node_filter_ast(allow, deny):
    """NodeFilter.ToAST (planner.go:83-116)"""
    a, d = len(allow), len(deny)
    if a == 0:
        if d == 0:
            return FALSE
        return deny[0] if d == 1 else ("and", list(deny))
    if a == 1:
        return allow[0] if d == 0 else ("and", list(deny) + [allow[0]])
    allow_f = ("or", list(allow))
    return allow_f if d == 0 else ("and", list(deny) + [allow_f])


class _Cond:
    """EvaluateCondition (planner.go:313-402) for one request"""

    def __init__(self, planner, principal, resource, aux_data, globals_, strict, errors, now_ns):
        self.p = planner
        self.principal, self.resource, self.aux, self.globals = principal, resource, aux_data, globals_ or {}
        self.strict, self.errors, self.now_ns = strict, errors, now_ns

    def condition(self, cond, consts, variables, dr_list):   # noqa: C901
        if cond is None:
            return TRUE
        kind = cond[0]
        if kind == "expr":
            return ("expr", self.expression(cond[1], consts, variables, dr_list))
        nodes = []
        for c in cond[1]:
            node = self.condition(c, consts, variables, dr_list)
            isc, bv = const_bool(node)
            if kind == "any":
                if isc:
                    if bv:
                        return TRUE
                    continue
                nodes.append(node)
            elif kind == "all":
                if isc:
                    if not bv:
                        return FALSE
                    continue
                nodes.append(node)
            else:   # none
                if isc:
                    if bv:
                        return FALSE
                    continue
                nodes.append(invert(node))
        if not nodes:
            return FALSE if kind == "any" else TRUE
        if len(nodes) == 1:
            return nodes[0]
        return ("or" if kind == "any" else "and", nodes)

    def expression(self, text, consts, variables, dr_list):
        tree = inline_variables(self.p.parse(text), variables)
        if dr_list is not None:
            tree = _replace_runtime_edr(tree, dr_list)
        env = request_env(self.principal, self.resource, self.aux, self.globals, consts)
        try:
            pe = Partial(env, self.now_ns)
            r = pe.pe(tree)
            if r[0] == "r":
                r = pe.process_root(r[1])
        except CelEvalError as e:
            self.errors.append({"expr": text, "message": str(e)})
            if self.strict:
                raise StrictEvaluationError(text)
            return ("lit", "bool", False)
        if r[0] == "k":
            return ("lit", "bool", r[1] is True)
        return r[1]

    def derived_roles_list(self, kind, version, scope, all_roles):
        """The value of runtime.effectiveDerivedRoles at a scope: the names of the derived roles whose parent roles the principal has,
        each under its condition (plan.go:143-190, planner.MkDerivedRolesList)."""
        drs = self.p.rt["policy_derived_roles"].get(namer.resource_policy_fqn(kind, version, scope)) or {}
        items, failed = [], None
        for name in sorted(drs):
            dr = drs[name]
            if not (set(dr["parent_roles"]) & all_roles) and "*" not in dr["parent_roles"]:
                continue
            # (inside a definition the list is still empty: plan.go:142, 161)
            try:
                node = self.condition(dr["condition"], dr["constants"], self.p.variable_exprs(dr["ordered_variables"]), [])
            except StrictEvaluationError as e:
                # plan.go:157-176 drListErr: a definition that fails under strict evaluation spoils the LIST, not the action - the
                # error surfaces only where a condition really reads runtime.effectiveDerivedRoles (_replace_runtime_edr)
                failed = failed or e
                continue
            items.append((name, node))
        return _DrListError(failed) if failed is not None else items


class _DrListError:
    """A scope's derived-role list that could not be built: the strict-evaluation error of one of its definitions, kept until read"""

    def __init__(self, err):
        self.err = err


def _replace_runtime_edr(tree, dr_list):
    from .partial import substitute

    def repl(x):
        if x[0] == "select" and x[1] == ("ident", "runtime") and x[2] in ("effectiveDerivedRoles", "effective_derived_roles"):
            if isinstance(dr_list, _DrListError):
                raise dr_list.err
            return _dr_list_expr(dr_list)
        return None
    return substitute(tree, repl)


def _dr_list_expr(items):
    """MkDerivedRolesList (planner.go:844-893): (cond ? [name] : []) per role, added up right to left"""
    parts = [("tern", _node_to_cel(node), ("list", (("lit", "string", name),)), ("list", ())) for name, node in items]
    if not parts:
        return ("list", ())
    out = parts[-1]
    for p in reversed(parts[:-1]):
        out = ("bin", "+", p, out)
    return out


def _node_to_cel(node):
    if node[0] == "expr":
        return node[1]
    kids = [_node_to_cel(c) for c in node[1]]
    if node[0] == "not":
        return ("not", kids[0])
    out = kids[0]
    for k in kids[1:]:
        out = (node[0], out, k)
    return out


def plan_resources_response(planner, request: dict, **params):
    """svc.PlanResources (internal/svc/cerbos_svc.go:53-118) around Planner.plan: a PlanResourcesRequest (its auxData already the engine's
    AuxData: the claims of the verified JWT) -> the PlanResourcesResponse - requestId, action | actions, resourceKind, policyVersion, filter
    and, with includeMeta, meta {filterDebug, matchedScope | matchedScopes}."""
    one = request.get("action") or ""
    actions = [one] if one else list(request.get("actions") or [])
    out = planner.plan(dict(request, actions=actions), **params)
    res = request.get("resource") or {}
    resp = {"requestId": request.get("requestId", ""), "resourceKind": res.get("kind", ""), "policyVersion": res.get("policyVersion", ""),
            "filter": out["filter"]}
    if request.get("includeMeta"):
        resp["meta"] = {"filterDebug": out["filterDebug"]}
        if one:
            resp["meta"]["matchedScope"] = out["matchedScopes"].get(one, "")
        else:
            resp["meta"]["matchedScopes"] = out["matchedScopes"]
    if one:
        resp["action"] = one
    else:
        resp["actions"] = actions
    return resp
