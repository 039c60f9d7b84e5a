"""Decoding of the trace pass's log (``cbh_trace_batch``, include/cerbos_hip.h) into what the reference attaches to a
``CheckOutput`` beyond the effects: ``evaluation_errors`` (evaluator/cel_errors.go:48-118: deduplicated, sorted
(expression, message) pairs) and ``outputs`` (check.go:383-411, 776-807: one ``OutputEntry`` per visit of a rule with an
output expression, in the order check.go's loops reach them).

Nothing here evaluates anything: the device reports WHICH expression failed and with WHICH error (a code and a detail),
or WHAT VALUE an output expression produced (a tag and 64 bits that may point into the caller's own batch); this module
turns codes into cel-go's message texts and values into JSON.  Errors the device does not classify
(``CBH_ERR_OTHER``) and values it cannot build leave the input marked incomplete - never a guessed text."""
import struct

import numpy as np

from .flatten import HEAP_BATCH, RQ_ACT_CNT, RQ_ACT_OFF

TR_ERROR, TR_OUTPUT, TR_OUTPUT_ERROR, TR_INCOMPLETE, TR_OUTPUT_ELEMENT = 1, 2, 3, 4, 5
HEAP_LOCAL = 3


class _LocalList:
    """A list the program built in its lane's arena: the elements come in CBH_TR_OUTPUT_ELEMENT records."""

    def __init__(self, n):
        self.n = n
TR_DRFAIL = 32   # cbh_check_wave.h CBH_TR_DRFAIL
(ERR_OTHER, ERR_NO_SUCH_KEY, ERR_ATTR_MISSING, ERR_NO_SUCH_OVERLOAD, ERR_UNDEFINED_FIELD, ERR_DIV_BY_ZERO, ERR_MOD_BY_ZERO,
 ERR_INT_OVERFLOW, ERR_UINT_OVERFLOW, ERR_EDR_FAILED) = range(10)
T_NULL, T_BOOL, T_INT, T_UINT, T_DOUBLE, T_STRING, T_LIST, T_MAP, T_TIMESTAMP, T_DURATION = range(10)
T_EDRSET = 11   # include/cerbos_hip.h CBH_T_EDRSET
HEAP_TABLE, HEAP_ROLES = 0, 2
RECORD_WORDS = 8

_FIXED = {ERR_NO_SUCH_OVERLOAD: "no such overload", ERR_DIV_BY_ZERO: "division by zero", ERR_MOD_BY_ZERO: "modulus by zero",
          ERR_INT_OVERFLOW: "integer overflow", ERR_UINT_OVERFLOW: "unsigned integer overflow"}


class _Incomplete(Exception):
    pass


def request_roots(inp, globals_=None):
    """The maps the attribute columns are paths into (flatten.py)."""
    p, res = inp["principal"], inp["resource"]
    aux = inp.get("auxData") or {}
    return {"P": p.get("attr") or {}, "R": res.get("attr") or {}, "J": aux.get("jwt") or {},
            "S": {k: {"claims": (j or {}).get("claims") or {}} for k, j in (aux.get("jwts") or {}).items()}, "G": globals_ or {}}


class TraceDecoder:
    def __init__(self, lt, batch, inputs, globals_=None):
        self.lt, self.batch, self.inputs, self.globals = lt, batch, inputs, globals_
        self.K = len(lt.strings)
        self._local = None

    def string(self, sid):
        if sid < self.K:
            return self.lt.strings[sid]
        if self._local is None:
            b = self.batch
            data = b.str_bytes.tobytes()
            self._local = [data[b.str_off[i]:b.str_off[i + 1]].decode("utf-8") for i in range(b.n_strings)]
        return self._local[sid - self.K]

    def message(self, code_word, inp):
        """cel-go's text for an error code (include/cerbos_hip.h CBH_ERR_*)."""
        code, detail = code_word & 0xFF, code_word >> 8
        if code == ERR_EDR_FAILED:   # check.go:593-610
            return "failed to compute effective derived roles [%s]" % ", ".join(
                sorted(n for i, n in enumerate(self.lt.dr_names) if (detail >> i) & 1))
        if code in _FIXED:
            return _FIXED[code]
        if code == ERR_NO_SUCH_KEY:
            return "no such key: %s" % self.string(detail)
        if code == ERR_UNDEFINED_FIELD:
            return "undefined field '%s'" % self.lt.trace_strings[detail]
        if code == ERR_ATTR_MISSING:
            # the column's path did not resolve in this input: which step failed decides the text (a missing key, or a
            # select below something that is not a map)
            root, keys = self.lt.columns[detail]
            cur = request_roots(inp, self.globals)[root]
            for key in keys:
                if not isinstance(cur, dict):
                    return "no such overload"
                if key not in cur:
                    return "no such key: %s" % key
                cur = cur[key]
        raise _Incomplete()

    def value(self, tag, v, top=False):
        """A device value as a Python value that keeps its CEL type: None, bool, int (int / uint), float (double), str,
        list, dict."""
        if tag == T_NULL:
            return None
        if tag == T_BOOL:
            return bool(v)
        if tag == T_INT:
            return v - (1 << 64) if v >> 63 else v
        if tag == T_UINT:
            return v
        if tag == T_DOUBLE:
            return struct.unpack("<d", struct.pack("<Q", v))[0]
        if tag == T_STRING:
            return self.string(v & 0xFFFFFFFF)
        if tag == T_EDRSET:   # runtime.effectiveDerivedRoles as a value: the mask of the scope's derived roles -> their names, sorted (check.go:593-610)
            return sorted(n for i, n in enumerate(self.lt.dr_names) if (v >> i) & 1)
        if tag in (T_LIST, T_MAP):
            sel, off, n = v >> 62, (v >> 32) & 0x3FFFFFFF, v & 0xFFFFFFFF
            if sel == HEAP_ROLES:
                return [self.string(int(x)) for x in self.batch.roles[off:off + n]]
            if sel == HEAP_BATCH:
                tags, vals = self.batch.heap_tag, self.batch.heap_val
            elif sel == HEAP_TABLE:
                tags, vals = self.lt.theap
            elif sel == HEAP_LOCAL and tag == T_LIST and top:
                return _LocalList(n)
            else:
                raise _Incomplete()
            if tag == T_LIST:
                return [self.value(int(tags[off + i]), int(vals[off + i])) for i in range(n)]
            out = {}
            for i in range(n):
                out[self.value(int(tags[off + 2 * i]), int(vals[off + 2 * i]))] = self.value(int(tags[off + 2 * i + 1]), int(vals[off + 2 * i + 1]))
            return out
        raise _Incomplete()   # timestamps / durations as output values: formatting left to the caller's engine

    def decode(self, records, count, status=None):
        """-> per INPUT of the traced batch: {"evaluationErrors": [...], "outputs": [...], "incomplete": set of "errors" /
        "outputs" the device could not name all of}.
        ``records``: uint32[capacity][8]; ``status``: the trace launch's per-tuple status (device order)."""
        b, lt = self.batch, self.lt
        n_in = len(self.inputs)
        errs = [set() for _ in range(n_in)]
        visits = [[] for _ in range(n_in)]
        parts = [{} for _ in range(n_in)]   # per input: visit -> the parts of its output expression logged so far
        incomplete = [set() for _ in range(n_in)]
        if count > len(records):
            raise RuntimeError("trace log overflow: %d records, room for %d" % (count, len(records)))
        rp = b.req_perm
        if status is not None:   # an operation outside the device subset inside a trace program
            bad = np.nonzero(np.asarray(status) == 2)[0]
            if bad.size:
                off, cnt = b.req_u32[RQ_ACT_OFF].astype(np.int64), b.req_u32[RQ_ACT_CNT].astype(np.int64)
                owner = np.searchsorted(off + cnt, bad, side="right")
                for r in owner:
                    incomplete[self._input_of(int(r))].update(("errors", "outputs"))
        for rec in records[:count]:
            r, w1, w2, w3 = int(rec[0]), int(rec[1]), int(rec[2]), int(rec[3])
            kind = w1 & 0xF
            i_in = self._input_of(r)
            inp = self.inputs[i_in]
            try:
                if kind == TR_INCOMPLETE:
                    incomplete[i_in].add("outputs")
                elif kind == TR_ERROR:
                    errs[i_in].add((lt.trace_strings[w2], self.message(int(rec[4]) | (int(rec[5]) << 32), inp)))
                elif kind in (TR_OUTPUT, TR_OUTPUT_ERROR, TR_OUTPUT_ELEMENT):
                    # one record per computed part of the output expression: collected per visit, assembled below
                    rule = (w3 >> 8) if kind != TR_OUTPUT_ERROR else int(rec[4])
                    pre = r if rp is None else int(rp[r])   # the request's index before the routing sort
                    visit = (pre, (w1 >> 4) & 1, (w1 >> 12) & 0xFF, w1 >> 20, rule)
                    part = (w1 >> 6) & 63
                    g = parts[i_in].setdefault(visit, {"src": w2, "mask": 0, "drfail": bool(w1 & TR_DRFAIL), "parts": {}, "elems": {}})
                    if kind == TR_OUTPUT_ELEMENT:
                        g["elems"].setdefault(part, {})[int(rec[6])] = self.value(w3 & 0xFF, int(rec[4]) | (int(rec[5]) << 32))
                        continue
                    g["mask"] = int(rec[6]) | (int(rec[7]) << 32)
                    if kind == TR_OUTPUT:
                        g["parts"][part] = (True, self.value(w3 & 0xFF, int(rec[4]) | (int(rec[5]) << 32), top=True))
                    else:
                        g["parts"][part] = (False, self.message(w3 | (int(rec[5]) << 32), inp))
            except _Incomplete:
                incomplete[i_in].add("errors" if kind == TR_ERROR else "outputs")
        for i_in in range(n_in):
            for (pre, pas, ri, site, rule), g in parts[i_in].items():
                tmpl, n_holes = lt.trace_templates[rule]
                entry = {"src": lt.trace_strings[g["src"]]}
                try:
                    if len(g["parts"]) != n_holes:
                        raise _Incomplete()
                    failed = [j for j in sorted(g["parts"]) if not g["parts"][j][0]]
                    if failed:   # the first part to fail in evaluation order is the expression's error
                        entry["error"] = g["parts"][failed[0]][1]
                    else:
                        holes = []
                        for j in range(n_holes):
                            v = g["parts"][j][1]
                            if isinstance(v, _LocalList):   # its elements were logged one by one
                                el = g["elems"].get(j, {})
                                if sorted(el) != list(range(v.n)):
                                    raise _Incomplete()
                                v = [el[k] for k in range(v.n)]
                            holes.append(v)
                        entry["val"] = _json(_assemble(tmpl, holes))
                except _Incomplete:
                    incomplete[i_in].add("outputs")
                    continue
                acts = b.vreq_actions[pre]
                for k in range(len(acts)):
                    if (g["mask"] >> k) & 1:
                        # order of check.go's loops: action, policy kind (principal first), role, scope / rule
                        visits[i_in].append(((pre, k, pas, ri, site), g["drfail"], dict(entry, action=acts[k]), rule & 0x7FFFFF))
        out = []
        for i in range(n_in):
            outs, seen_drfail = [], set()
            for _key, drfail, entry, rule in sorted(visits[i], key=lambda v: v[0]):
                if drfail:
                    # the first visit of a rule whose derived-role condition fails emits nothing (check.go:343-347 caches
                    # "false" under the evaluation key and moves on); later visits find the cached outcome and emit conditionNotMet
                    if rule not in seen_drfail:
                        seen_drfail.add(rule)
                        continue
                outs.append(entry)
            out.append({"evaluationErrors": [{"celError": {"expression": e, "message": m}} for e, m in sorted(errs[i])],
                        "outputs": outs, "incomplete": incomplete[i]})
        return out

    def _input_of(self, r):
        b = self.batch
        pre = r if b.req_perm is None else int(b.req_perm[r])
        return int(b.vreq_input[pre])


def _assemble(t, holes):
    """The value of an output expression from its template (celc.py _output_template) and the values of its holes."""
    k = t[0]
    if k == "hole":
        return holes[t[1]]
    if k == "const":
        return t[1]
    if k == "list":
        return [_assemble(e, holes) for e in t[1]]
    if k == "map":
        out = {}
        for kt, vt in t[1]:
            key = _assemble(kt, holes)
            if isinstance(key, (list, dict)) or key is None or key in out:
                raise _Incomplete()   # "unsupported key type" / a repeated key: errors of the literal, left to the caller's engine
            out[key] = _assemble(vt, holes)
        return out
    return _format(t[1], [_assemble(e, holes) for e in t[2]])


def _format(fmt, args):
    """cel-go ext.Strings `format` (strings.go): the clauses that need no decision about precision or radix - %s, %d, %% -
    anything else is left to the caller's engine."""
    out, i, ai = [], 0, 0
    while i < len(fmt):
        c = fmt[i]
        i += 1
        if c != "%":
            out.append(c)
            continue
        if i >= len(fmt):
            raise _Incomplete()
        spec = fmt[i]
        i += 1
        if spec == "%":
            out.append("%")
            continue
        if ai >= len(args) or spec not in "sd":
            raise _Incomplete()
        a = args[ai]
        ai += 1
        if spec == "d":
            if isinstance(a, bool) or not isinstance(a, (int, float)) or (isinstance(a, float) and a != int(a)):
                raise _Incomplete()   # "decimal clause can only be used on integers"
            out.append(str(int(a)))
        else:
            out.append(_format_value(a))
    return "".join(out)


def _format_value(v, nested=False):
    """%s of a value (cel-go strings.go formatString and the per-type formatters)."""
    if v is None:
        return "null"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, str):
        return '"%s"' % v if nested else v
    if isinstance(v, int):
        return str(v)
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            raise _Incomplete()
        return str(int(v)) if v == int(v) and abs(v) < 1e21 else repr(v)
    if isinstance(v, list):
        return "[" + ", ".join(_format_value(x, True) for x in v) + "]"
    if isinstance(v, dict):
        return "{" + ", ".join("%s: %s" % (_format_value(k, True), _format_value(x, True))
                               for k, x in sorted(v.items(), key=lambda kv: str(kv[0]))) + "}"
    raise _Incomplete()


def _json(v):
    """structpb JSON of a value (check.go:789-807): every number a double, map keys their text."""
    if isinstance(v, bool) or v is None or isinstance(v, str):
        return v
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, list):
        return [_json(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _json(x) for k, x in v.items()}
    raise _Incomplete()


def _key_text(k):
    if isinstance(k, bool):
        return "true" if k else "false"
    if isinstance(k, float) and k == int(k):
        return str(int(k))
    return str(k)
