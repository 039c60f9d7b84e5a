"""The planner's node tree -> the filter of a PlanResourcesOutput (internal/ruletable/planner/ast.go: convert / buildExpr, the lambda
forms of the comprehension macros, normaliseFilter) and the AND of the per-action filters (merge.go).

Node tree (what cerbos_amd/plan/planner.py builds, planner.go's PlanResourcesAst_Node): ("and" | "or" | "not", [nodes]) logical
operations and ("expr", cel_ast) leaves, a leaf's CEL tree in the parser's tuple form.  Filter operands are JSON-shaped dicts:
{"expression": {"operator": .., "operands": [..]}} | {"variable": "a.b"} | {"value": ..}."""
from __future__ import annotations

_BIN = {"==": "eq", "!=": "ne", "<": "lt", "<=": "le", ">": "gt", ">=": "ge", "in": "in", "+": "add", "-": "sub", "*": "mult", "/": "div", "%": "mod"}
_ABBREV = {"R": "request.resource", "P": "request.principal", "G": "globals", "C": "constants", "V": "variables"}
# macro -> operator (lambda.go:60-120)
_MACRO = {"all": "all", "exists": "exists", "exists_one": "exists_one", "existsOne": "exists_one", "map": "map", "filter": "filter",
          "transformList": "transformList", "transformMap": "transformMap", "transformMapEntry": "transformMapEntry", "sortBy": "sortBy"}
_ON_STRUCT = ("transformMap", "transformMapEntry", "transformList")   # canOperateOnStruct (ast.go:875-877): the others range over a struct's keys


def expr(op, *operands):
    return {"expression": {"operator": op, "operands": list(operands)}}


def _value(n):
    k = n[1]
    if k in ("int", "uint", "double"):
        return float(n[2]) if not isinstance(n[2], bool) else n[2]
    if k == "bytes":
        import base64
        return base64.b64encode(bytes(n[2])).decode("ascii")
    return n[2]


def _is_const(n):
    return n[0] == "lit"


def _select_chain(n):
    names = []
    while n[0] == "select":
        names.append(n[2])
        n = n[1]
    if n[0] == "ident":
        names.append(n[1])
        return ".".join(reversed(names))
    return None


def _map_keys_list(n):
    return ("list", tuple(k for k, _ in n[1]))


def build(n, parent=None):   # noqa: C901
    """buildExprImpl (ast.go:363-514)"""
    k = n[0]
    if k == "lit":
        return {"value": _value(n)}
    if k == "ident":
        return {"variable": n[1]}
    if k == "has":
        return {"value": True}
    if k == "select":
        name = _select_chain(n)
        if name is not None:
            return {"variable": name}
        return expr("get-field", build(n[1], n), {"variable": n[2]})
    if k == "list":
        if all(_is_const(e) for e in n[1]):
            return {"value": [_value(e) for e in n[1]]}
        return expr("list", *[build(e, n) for e in n[1]])
    if k == "map":
        if parent is not None and parent[0] == "bin" and parent[1] == "in" and parent[3] is n:
            return build(_map_keys_list(n), parent)
        ents = []
        for ke, ve in n[1]:
            ents.append(expr("set-field", build(ke, n), build(ve, n)))
        return expr("struct", *ents)
    if k == "bin":
        return expr(_BIN[n[1]], build(n[2], n), build(n[3], n))
    if k in ("and", "or"):
        return expr(k, build(n[1], n), build(n[2], n))
    if k == "not":
        return expr("not", build(n[1], n))
    if k == "neg":
        return expr("-_", build(n[1], n))
    if k == "tern":
        return expr("if", build(n[1], n), build(n[2], n), build(n[3], n))
    if k == "index":
        return expr("index", build(n[1], n), build(n[2], n))
    if k == "call":
        ops = ([build(n[2], n)] if n[2] is not None else []) + [build(a, n) for a in n[3]]
        return expr(n[1], *ops)
    if k == "comp":
        _, macro, target, vars_, args = n
        op = _MACRO[macro]
        if op == "map" and len(args) == 2:        # map(x, filter, transform): a filter inside (lambda.go: Conditional over a list)
            op = "transformList" if len(vars_) == 2 else "map"
        if op == "transformMap" and len(args) == 1 and macro == "transformMapEntry":
            op = "transformMapEntry"
        rng = target
        if rng[0] == "map" and op not in _ON_STRUCT:
            rng = _map_keys_list(rng)
        lam = [build(a, n) for a in args] + [{"variable": v} for v in vars_]
        return expr(op, build(rng, n), expr("lambda", *lam))
    if k == "bind":
        raise ValueError("cel.bind in a residual expression")
    raise ValueError("unsupported expression %r" % (k,))


def convert(node):
    """convert (ast.go:256-297)"""
    if node is None:
        return None
    k = node[0]
    if k == "expr":
        return build(node[1])
    return expr(k, *[convert(c) for c in node[1]])


# ---- normaliseFilter (ast.go:596-819)
def _as_bool(op):
    if op is not None and "value" in op and isinstance(op["value"], bool):
        return True, op["value"]
    return False, False


def _expand_abbrev(name):
    head, _, rest = name.partition(".")
    full = _ABBREV.get(head)
    if full is None:
        return name
    return full + ("." + rest if rest else "")


def _key(op):
    import json
    return json.dumps(op, sort_keys=True)


def normalise_operand(op):   # noqa: C901
    if op is None:
        return None
    if "variable" in op:
        return {"variable": _expand_abbrev(op["variable"])}
    if "value" in op:
        return op
    e = op["expression"]
    oper, operands = e["operator"], e["operands"]
    if oper == "in" and len(operands) == 2 and "value" in operands[1]:
        v = operands[1]["value"]
        if isinstance(v, dict):
            if len(v) == 0:
                return {"value": False}
            if len(v) == 1:
                oper, operands = "eq", [operands[0], {"value": next(iter(v))}]
        elif isinstance(v, list):
            if len(v) == 0:
                return {"value": False}
            if len(v) == 1:
                oper, operands = "eq", [operands[0], {"value": v[0]}]
        else:
            oper = "eq"
    logical = oper if oper in ("and", "or", "not") else ""
    seen = set() if logical in ("and", "or") else None
    out = []
    for o in operands:
        no = normalise_operand(o)
        if no is None:
            continue
        if logical:
            isb, bv = _as_bool(no)
            if isb:
                if logical == "and" and bv:
                    continue
                if logical == "or" and not bv:
                    continue
                if logical == "and":
                    return {"value": False}
                if logical == "or":
                    return {"value": True}
        if seen is not None:
            h = _key(o)
            if h in seen:
                continue
            seen.add(h)
        out.append(no)
    if logical:
        if len(out) == 0:
            if logical == "and":
                return {"value": True}
            if logical == "or":
                return {"value": False}
            return None
        if len(out) == 1:
            if logical in ("and", "or"):
                return out[0]
            isb, bv = _as_bool(out[0])
            if isb:
                return {"value": not bv}
    return {"expression": {"operator": oper, "operands": out}}


def to_filter(node):
    """ToFilter + normaliseFilter -> {"kind": .., "condition": ..}"""
    cond = normalise_operand(convert(node))
    if cond is None:
        return {"kind": "KIND_ALWAYS_ALLOWED"}
    isb, bv = _as_bool(cond)
    if isb:
        return {"kind": "KIND_ALWAYS_ALLOWED" if bv else "KIND_ALWAYS_DENIED"}
    return {"kind": "KIND_CONDITIONAL", "condition": cond}


def merge_with_and(filters):
    """merge.go:14-48: the call's filter = AND of its actions' conditional filters (equal ones once), denied if any action is."""
    conds = {}
    for f in filters:
        if f["kind"] == "KIND_ALWAYS_DENIED":
            return {"kind": "KIND_ALWAYS_DENIED"}
        if f["kind"] == "KIND_CONDITIONAL":
            conds[_key(f["condition"])] = f["condition"]
    if not conds:
        return {"kind": "KIND_ALWAYS_ALLOWED"}
    if len(conds) == 1:
        return {"kind": "KIND_CONDITIONAL", "condition": next(iter(conds.values()))}
    return {"kind": "KIND_CONDITIONAL", "condition": expr("and", *[conds[k] for k in sorted(conds)])}


def normalise_filter(f):
    """normaliseFilter (ast.go:596-620) on a filter given as data"""
    if f.get("kind") != "KIND_CONDITIONAL":
        return {"kind": f.get("kind")}
    cond = normalise_operand(f.get("condition")) if f.get("condition") else None
    if cond is None:
        return {"kind": "KIND_ALWAYS_ALLOWED"}
    isb, bv = _as_bool(cond)
    if isb:
        return {"kind": "KIND_ALWAYS_ALLOWED" if bv else "KIND_ALWAYS_DENIED"}
    return {"kind": "KIND_CONDITIONAL", "condition": cond}


def _json(v):
    """protojson of a google.protobuf.Value, compact"""
    import json
    if isinstance(v, float) and v == int(v) and abs(v) < 1e15:
        return json.dumps(int(v))
    if isinstance(v, list):
        return "[" + ",".join(_json(x) for x in v) + "]"
    if isinstance(v, dict):
        return "{" + ",".join(json.dumps(k) + ":" + _json(x) for k, x in v.items()) + "}"
    return json.dumps(v, ensure_ascii=False)


def operand_to_string(op):
    if op is None:
        return ""
    if "expression" in op:
        e = op["expression"]
        return "(" + e["operator"] + " " + " ".join(operand_to_string(o) for o in e.get("operands") or []) + ")"
    if "value" in op:
        return _json(op["value"])
    return op["variable"]


def filter_to_string(f):
    """FilterToString (ast.go:821-834): the filterDebug of an output"""
    if f["kind"] == "KIND_ALWAYS_ALLOWED":
        return "(true)"
    if f["kind"] == "KIND_ALWAYS_DENIED":
        return "(false)"
    return operand_to_string(f.get("condition"))
