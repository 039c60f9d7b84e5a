"""The two checks of cel-go's type checker that the reference's policy compiler tests pin (internal/compile/compile_test.go over
internal/test/testdata/compile: bad_variables.yaml, variables_index_lookup.yaml), on this package's AST:

* an identifier the environment does not declare - `undeclared reference to 'wat' (in container '')`.  The environment is
  internal/conditions/cel.go:42-53 (request, P, R, runtime, constants / C, variables / V, globals / G) plus what a macro binds;
* `V["x"]`, `C["x"]`, `G["x"]`: cerbos.Variables is an object type, it has fields and no index operator -
  `found no matching overload for '_[_]' applied to '(cerbos.Variables, string)'`.

* a field the message types of the environment do not have - `undefined field 'wat'` for `request.wat`, `runtime.wat`,
  `P.wat`, `request.aux_data.wat` ... (internal/conditions/types/registry_test.go TestJSONFields / TestRuntime pin the text;
  the messages are engine.proto:304-322, 397-422 Request / Request.Principal / Request.Resource / AuxData / AuxData.JWT / Runtime, every field
  under its proto name and its JSON name, registry.go:58-125).

Everything else cel-go's checker would reject (unknown functions, argument types, a field selected from a scalar) is NOT
checked: the lowering refuses what it cannot compile, but a policy the reference rejects for those reasons is accepted here."""
from . import parser

DECLARED = frozenset(("request", "P", "R", "runtime", "constants", "C", "variables", "V", "globals", "G"))
VARIABLES_TYPED = frozenset(("constants", "C", "variables", "V", "globals", "G"))
# type names are identifiers of the standard environment; google.* / cerbos.* start qualified message and enum names
_TYPES = frozenset(("bool", "bytes", "double", "int", "uint", "string", "list", "map", "null_type", "type", "optional_type", "google", "cerbos"))
# message types: field (proto or JSON name) -> the message type behind it, ("map", type) for a map of messages, None where the
# value is a scalar / list / map of values
_PRINCIPAL = {"id": None, "roles": None, "attr": None, "policy_version": None, "policyVersion": None, "scope": None}
_RESOURCE = {"kind": None, "id": None, "attr": None, "policy_version": None, "policyVersion": None, "scope": None}
_JWT = {"claims": None}
_AUX = {"jwt": None, "jwts": ("map", _JWT)}
_REQUEST = {"principal": _PRINCIPAL, "resource": _RESOURCE, "aux_data": _AUX, "auxData": _AUX}
_RUNTIME = {"effective_derived_roles": None, "effectiveDerivedRoles": None}
_MESSAGES = {"request": _REQUEST, "P": _PRINCIPAL, "R": _RESOURCE, "runtime": _RUNTIME}
_LIT_TYPE = {"null": "null", "bool": "bool", "int": "int", "uint": "uint", "double": "double", "string": "string", "bytes": "bytes"}


def issues(ast):
    """The messages, in source order of the offending nodes."""
    out = []
    _visit(ast, frozenset(), out)
    return out


def _message_type(n, bound):
    """The message type of the value of `n`, when it is one of the environment's (else None)."""
    if n[0] == "ident":
        return _MESSAGES.get(n[1]) if n[1] not in bound else None
    if n[0] in ("select", "index"):
        parent = _message_type(n[1], bound)
        if isinstance(parent, tuple):      # a map of messages: any key
            return parent[1]
        return parent.get(n[2]) if isinstance(parent, dict) and n[0] == "select" else None
    return None


def _visit(n, bound, out):   # noqa: C901
    k = n[0]
    if k == "lit":
        return
    if k == "ident":
        if n[1] not in bound and n[1] not in DECLARED and n[1] not in _TYPES:
            out.append("undeclared reference to '%s' (in container '')" % n[1])
        return
    if k in ("select", "has"):
        _visit(n[1], bound, out)
        msg = _message_type(n[1], bound)
        if isinstance(msg, dict) and n[2] not in msg:
            out.append("undefined field '%s'" % n[2])
    elif k == "index":
        _visit(n[1], bound, out)
        _visit(n[2], bound, out)
        if n[1][0] == "ident" and n[1][1] in VARIABLES_TYPED and n[1][1] not in bound and n[2][0] == "lit":
            out.append("found no matching overload for '_[_]' applied to '(cerbos.Variables, %s)'" % _LIT_TYPE[n[2][1]])
    elif k == "call":
        _, _name, target, args = n
        # `ns.fn(...)`: the target of a call may be the namespace of a function (sets, math, lists, base64, regex, optional, ip, cel ...)
        if target is not None and not (target[0] == "ident" and target[1] not in bound and target[1] not in DECLARED):
            _visit(target, bound, out)
        for a in args:
            _visit(a, bound, out)
    elif k == "list":
        for e in n[1]:
            _visit(e, bound, out)
    elif k == "map":
        for kk, vv in n[1]:
            _visit(kk, bound, out)
            _visit(vv, bound, out)
    elif k in ("not", "neg"):
        _visit(n[1], bound, out)
    elif k == "bin":
        _visit(n[2], bound, out)
        _visit(n[3], bound, out)
    elif k in ("and", "or"):
        _visit(n[1], bound, out)
        _visit(n[2], bound, out)
    elif k == "tern":
        for x in n[1:4]:
            _visit(x, bound, out)
    elif k == "comp":
        _, _kind, target, names, args = n
        _visit(target, bound, out)
        inner = bound | frozenset(names)
        for a in args:
            _visit(a, inner, out)
    elif k == "bind":
        _, name, init, body = n
        _visit(init, bound, out)
        _visit(body, bound | {name}, out)
    else:
        raise AssertionError("unknown AST node %r" % (k,))


def compile_issues(text):
    """conditions.Compile (cel.go:170-176) as far as it is restated: (ast or None, messages)."""
    try:
        ast = parser.parse(text)
    except parser.CELSyntaxError as err:
        return None, ["Syntax error: %s" % err]
    return ast, issues(ast)
