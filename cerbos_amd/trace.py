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

TR_ERROR, TR_OUTPUT, TR_OUTPUT_ERROR, TR_INCOMPLETE = 1, 2, 3, 4
TR_DRFAIL = 32   # cbh_check_wave.h CBH_TR_DRFAIL
(ERR_OTHER, ERR_NO_SUCH_KEY, ERR_ATTR_MISSING, ERR_NO_SUCH_OVERLOAD, ERR_UNDEFINED_FIELD, ERR_DIV_BY_ZERO, ERR_MOD_BY_ZERO,
 ERR_INT_OVERFLOW, ERR_UINT_OVERFLOW, ERR_EDR_FAILED) = range(10)
T_NULL, T_BOOL, T_INT, T_UINT, T_DOUBLE, T_STRING, T_LIST, T_MAP, T_TIMESTAMP, T_DURATION = range(10)
HEAP_TABLE, HEAP_ROLES = 0, 2
RECORD_WORDS = 8

_FIXED = {ERR_NO_SUCH_OVERLOAD: "no such overload", ERR_DIV_BY_ZERO: "division by zero", ERR_MOD_BY_ZERO: "modulus by zero",
          ERR_INT_OVERFLOW: "integer overflow", ERR_UINT_OVERFLOW: "unsigned integer overflow"}


class _Incomplete(Exception):
    pass


def request_roots(inp):
    """The maps the attribute columns are paths into (flatten.py)."""
    p, res = inp["principal"], inp["resource"]
    aux = inp.get("auxData") or {}
    return {"P": p.get("attr") or {}, "R": res.get("attr") or {}, "J": aux.get("jwt") or {},
            "S": {k: {"claims": (j or {}).get("claims") or {}} for k, j in (aux.get("jwts") or {}).items()}}


class TraceDecoder:
    def __init__(self, lt, batch, inputs):
        self.lt, self.batch, self.inputs = lt, batch, inputs
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
            cur = request_roots(inp)[root]
            for key in keys:
                if not isinstance(cur, dict):
                    return "no such overload"
                if key not in cur:
                    return "no such key: %s" % key
                cur = cur[key]
        raise _Incomplete()

    def value(self, tag, v):
        """structpb JSON of an output value (check.go:789-807)."""
        if tag == T_NULL:
            return None
        if tag == T_BOOL:
            return bool(v)
        if tag == T_INT:
            return float(v - (1 << 64) if v >> 63 else v)
        if tag == T_UINT:
            return float(v)
        if tag == T_DOUBLE:
            return struct.unpack("<d", struct.pack("<Q", v))[0]
        if tag == T_STRING:
            return self.string(v & 0xFFFFFFFF)
        if tag in (T_LIST, T_MAP):
            sel, off, n = v >> 62, (v >> 32) & 0x3FFFFFFF, v & 0xFFFFFFFF
            if sel == HEAP_ROLES:
                return [self.string(int(x)) for x in self.batch.roles[off:off + n]]
            if sel == HEAP_BATCH:
                tags, vals = self.batch.heap_tag, self.batch.heap_val
            elif sel == HEAP_TABLE:
                tags, vals = self.lt.theap
            else:
                raise _Incomplete()
            if tag == T_LIST:
                return [self.value(int(tags[off + i]), int(vals[off + i])) for i in range(n)]
            out = {}
            for i in range(n):
                k = self.value(int(tags[off + 2 * i]), int(vals[off + 2 * i]))
                out[k if isinstance(k, str) else _key_text(k)] = self.value(int(tags[off + 2 * i + 1]), int(vals[off + 2 * i + 1]))
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
                elif kind in (TR_OUTPUT, TR_OUTPUT_ERROR):
                    mask = int(rec[6]) | (int(rec[7]) << 32)
                    entry = {"src": lt.trace_strings[w2]}
                    if kind == TR_OUTPUT:
                        rule = w3 >> 8
                        entry["val"] = self.value(w3 & 0xFF, int(rec[4]) | (int(rec[5]) << 32))
                    else:
                        rule = int(rec[4])
                        entry["error"] = self.message(w3 | (int(rec[5]) << 32), inp)
                    pre = r if rp is None else int(rp[r])   # the request's index before the routing sort
                    acts = b.vreq_actions[pre]
                    for k in range(len(acts)):
                        if (mask >> k) & 1:
                            # order of check.go's loops: action, policy kind (principal first), role, scope / rule
                            visits[i_in].append(((pre, k, (w1 >> 4) & 1, (w1 >> 12) & 0xFF, w1 >> 20), bool(w1 & TR_DRFAIL),
                                                 dict(entry, action=acts[k]), rule))
            except _Incomplete:
                incomplete[i_in].add("errors" if kind == TR_ERROR else "outputs")
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


def _key_text(k):
    if isinstance(k, bool):
        return "true" if k else "false"
    if isinstance(k, float) and k == int(k):
        return str(int(k))
    return str(k)
