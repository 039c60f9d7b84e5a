"""CEL condition trees -> device bytecode (the "CEL bytecode tape" of the table image).

Input: the condition tuple trees of rule-table rows (``cerbos_amd.policy.compile``) and
the row's params (constants + ordered variables).  Policy variables / constants / globals
are inlined into the expression (CEL is side-effect free, so ``V.x`` evaluated at its use
is equivalent to the reference evaluating it once per request - check.go:651-677 - as far
as effects go; the evaluation_errors *text* differs and is not produced by the device).

Instruction encoding and operand-stack discipline: see cerbos_amd/csrc/cbh_blob.h (CbhOp)
and cbh_vm.h (run_program).  Anything outside the device subset compiles to
OP_UNSUPPORTED: the table still loads, and a tuple that actually executes such an
instruction is reported with status CBH_ST_UNSUPPORTED (never a silently wrong effect).
"""
from __future__ import annotations

import re

import struct

from ..cel import fold as celfold
from ..cel import parser as celparser
from . import regex

# keep in sync with cbh_blob.h
(OP_RET, OP_CONST, OP_COL, OP_HASCOL, OP_REQSTR, OP_ROLES, OP_SELECT, OP_HASSEL, OP_INDEX,
 OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_IN, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD,
 OP_NEG, OP_NOT, OP_JF, OP_JT, OP_AND, OP_OR, OP_JTERN, OP_JMP, OP_POP, OP_LEAF, OP_SIZE,
 OP_STARTSWITH, OP_ENDSWITH, OP_CONTAINS, OP_TIMESTAMP, OP_DURATION, OP_TIMESINCE, OP_NOW,
 OP_EDRHAS, OP_LOCAL, OP_ITER_BEGIN, OP_ITER_NEXT, OP_ITER_ACC, OP_ITER_END, OP_TOINT,
 OP_TODOUBLE, OP_TOSTRING_UNSUPPORTED, OP_INIPRANGE, OP_UNSUPPORTED, OP_TS_GETTER,
 OP_HASINTERSECTION, OP_ISSUBSET, OP_LEAF_BIN, OP_TERN, OP_TREE_BEGIN, OP_TREE_ACC,
 OP_TREE_END, OP_HIER, OP_MATCHES, OP_INDEXOF, OP_STREQ_CASE, OP_VARSCOPE, OP_OUT, OP_LISTOP, OP_LISTFN, OP_STRCAT,
 OP_STRCASE, OP_IPFN, OP_STRVIEW, OP_STRREPLACE, OP_HIERCOMMON, OP_EDREQ, OP_EDRVAL) = range(74)

TREE_KINDS = {"all": 0, "any": 1, "none": 2}
COND_LEAF = 0x80000000
COND_LEAFTREE = 0x40000000
COND_PC_MASK = 0x3FFFFFFF


def _is_leaf_tree(words):
    """True if the program is only TREE_* structure around fused leaves (no operand stack needed)."""
    i, n = 0, len(words)
    while i < n:
        op = words[i] & 0xFF
        if op == OP_LEAF_BIN:
            i += 3
        elif op in (OP_TREE_BEGIN, OP_TREE_ACC, OP_TREE_END):
            i += 1
        elif op == OP_RET and i == n - 1:
            return True
        else:
            return False
    return False

T_NULL, T_BOOL, T_INT, T_UINT, T_DOUBLE, T_STRING, T_LIST, T_MAP, T_TIMESTAMP, T_DURATION = range(10)
T_ABSENT, T_ERR = 0xF0, 0xFF

HEAP_TABLE, HEAP_BATCH, HEAP_ROLES = 0, 1, 2

# request string fields (cbh_req_field)
RQ_PRINCIPAL_ID, RQ_S_RESOURCE_ID, RQ_S_KIND = 0, 10, 11
RQ_S_P_SCOPE, RQ_S_R_SCOPE, RQ_S_P_VERSION, RQ_S_R_VERSION = 12, 13, 14, 15

IT_ALL, IT_EXISTS, IT_EXISTS_ONE, IT_FILTER, IT_MAP, IT_MAP_FILTER = 0, 1, 2, 3, 4, 5

MAX_STACK = 10
TREE_STRIP_MAX = 8   # leaves of a condition tree the flat kernel evaluates inline (cbh_blob.h CBH_TREE_STRIP_MAX)
CACHE_COLS = 16      # CBH_CACHE_COLS (cbh_vm.h): columns the decision kernel parks in LDS
MAX_LOCALS = 4
MAX_ITERS = 2

_BINOPS = {"==": OP_EQ, "!=": OP_NE, "<": OP_LT, "<=": OP_LE, ">": OP_GT, ">=": OP_GE, "in": OP_IN,
           "+": OP_ADD, "-": OP_SUB, "*": OP_MUL, "/": OP_DIV, "%": OP_MOD}

_P_FIELDS = {"id": RQ_PRINCIPAL_ID, "scope": RQ_S_P_SCOPE, "policyVersion": RQ_S_P_VERSION,
             "policy_version": RQ_S_P_VERSION}
_R_FIELDS = {"id": RQ_S_RESOURCE_ID, "kind": RQ_S_KIND, "scope": RQ_S_R_SCOPE,
             "policyVersion": RQ_S_R_VERSION, "policy_version": RQ_S_R_VERSION}


# opcodes that never look inside a string: they move values, compare interned ids / numbers, or steer control flow.
# A table whose programs use only these needs no batch-local string bytes on the device (cbh_engine.hip skips
# their upload); orderings are decided per use (OP_LEAF_BIN in _fused_leaf, OP_LT.. here count as content reads).
_ID_ONLY_OPS = frozenset([OP_RET, OP_CONST, OP_COL, OP_HASCOL, OP_REQSTR, OP_ROLES, OP_SELECT, OP_HASSEL, OP_INDEX, OP_EQ, OP_NE,
                          OP_IN, OP_NOT, OP_JF, OP_JT, OP_AND, OP_OR, OP_JTERN, OP_JMP, OP_POP, OP_LEAF, OP_EDRHAS, OP_LOCAL,
                          OP_ITER_BEGIN, OP_ITER_NEXT, OP_ITER_ACC, OP_ITER_END, OP_HASINTERSECTION, OP_ISSUBSET, OP_LEAF_BIN,
                          OP_TERN, OP_TREE_BEGIN, OP_TREE_ACC, OP_TREE_END, OP_UNSUPPORTED, OP_TS_GETTER, OP_VARSCOPE, OP_OUT, OP_LISTOP, OP_LISTFN, OP_EDREQ, OP_EDRVAL])


class LoweringError(ValueError):
    pass


def f64_bits(x: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def cont_payload(sel: int, off: int, length: int) -> int:
    return (sel << 62) | (off << 32) | length


def value_to_ast(v):
    """A constant / global value (YAML/JSON) as a CEL literal AST; numbers are doubles
    because they travel as google.protobuf.Value (structpb)."""
    if v is None:
        return ("lit", "null", None)
    if isinstance(v, bool):
        return ("lit", "bool", v)
    if isinstance(v, (int, float)):
        return ("lit", "double", float(v))
    if isinstance(v, str):
        return ("lit", "string", v)
    if isinstance(v, (list, tuple)):
        return ("list", tuple(value_to_ast(x) for x in v))
    if isinstance(v, dict):
        return ("map", tuple((("lit", "string", str(k)), value_to_ast(x)) for k, x in v.items()))
    raise LoweringError("unsupported constant value %r" % (v,))


def _subst(ast, fn):
    """Bottom-up rewrite."""
    k = ast[0]
    if k in ("lit", "ident"):
        return fn(ast)
    if k in ("select", "has"):
        return fn((k, _subst(ast[1], fn), ast[2]))
    if k == "index":
        return fn((k, _subst(ast[1], fn), _subst(ast[2], fn)))
    if k == "call":
        tgt = None if ast[2] is None else _subst(ast[2], fn)
        return fn((k, ast[1], tgt, tuple(_subst(a, fn) for a in ast[3])))
    if k == "list":
        return fn((k, tuple(_subst(a, fn) for a in ast[1])))
    if k == "map":
        return fn((k, tuple((_subst(a, fn), _subst(b, fn)) for a, b in ast[1])))
    if k in ("not", "neg"):
        return fn((k, _subst(ast[1], fn)))
    if k == "bin":
        return fn((k, ast[1], _subst(ast[2], fn), _subst(ast[3], fn)))
    if k in ("and", "or"):
        return fn((k, _subst(ast[1], fn), _subst(ast[2], fn)))
    if k == "tern":
        return fn((k, _subst(ast[1], fn), _subst(ast[2], fn), _subst(ast[3], fn)))
    if k == "comp":
        return fn((k, ast[1], _subst(ast[2], fn), ast[3], tuple(_subst(a, fn) for a in ast[4])))
    if k == "bind":
        return fn((k, ast[1], _subst(ast[2], fn), _subst(ast[3], fn)))
    return ast


class _PerCallGlobals(dict):
    """Marker: `G.x` / `globals.x` is NOT folded into the image; it compiles to a column of root "G" that every flattener fills
    from the globals of the CALL (evaluator.EvalParams.Globals, internal/evaluator/evaluator.go:52-57, 98-106) - one image
    answers any globals."""


PER_CALL_GLOBALS = _PerCallGlobals()


class Params:
    """Constants + variables visible to one condition (RuleRow.Params)."""

    def __init__(self, constants=None, ordered_variables=None, globals_=None, trace=False, null_on_error=False):
        self.constants = dict(constants or {})
        self.ordered_variables = list(ordered_variables or [])
        self.variables = {n: t for n, t in self.ordered_variables}
        self.globals = globals_ if isinstance(globals_, _PerCallGlobals) else dict(globals_ or {})
        # trace programs (ProgramBuilder.trace_*): an inlined variable stays recognisable - ("varscope", name, body) - because
        # the reference evaluates it on its own and a failure there reads differently at the place of use (check.go:651-677)
        self.trace = trace
        # the variables of a derived-role definition (evaluateVariables, check.go:612-633): a failing one is null, not unset
        self.null_on_error = null_on_error
        self._inlined = {}

    def key(self):
        return (tuple(sorted((k, repr(v)) for k, v in self.constants.items())),
                tuple(self.ordered_variables), self.trace, self.null_on_error)

    def inline(self, ast, depth=0):
        if depth > 32:
            raise LoweringError("variable definitions nest too deeply")

        def fn(n):
            if n[0] == "select" and n[1][0] == "ident":
                base, name = n[1][1], n[2]
                if base in ("V", "variables"):
                    if name not in self.variables:
                        return ("call", "__unsupported__", None, ())
                    if name not in self._inlined:
                        self._inlined[name] = self.inline(celparser.parse(self.variables[name]), depth + 1)
                    if self.trace or self.null_on_error:
                        return ("varscope", name, self._inlined[name], 1 if self.null_on_error else 0)
                    return self._inlined[name]
                if base in ("C", "constants"):
                    if name not in self.constants:
                        return ("call", "__unsupported__", None, ())
                    return value_to_ast(self.constants[name])
                if base in ("G", "globals"):
                    if isinstance(self.globals, _PerCallGlobals):
                        return n   # a column read (CelCompiler._path): the call's globals, not the lowering's
                    if name not in self.globals:
                        if self.trace:
                            return ("call", "__undef__", None, (("lit", "string", name),))
                        return ("call", "__error__", None, ())   # undefined field -> CEL error
                    return value_to_ast(self.globals[name])
            return n

        out = _subst(ast, fn)
        # constant sub-expressions are computed here, once, instead of on every request (cel/fold.py)
        return celfold.fold(out) if depth == 0 else out


class ProgramBuilder:
    """Accumulates code, constants, table heap, strings and the column schema for a table."""

    def __init__(self, intern_string, globals_=None):
        self.sid = intern_string           # str -> table string id
        self.per_call_globals = isinstance(globals_, _PerCallGlobals)
        self.globals = dict(globals_ or {})
        self.code = []
        self.const_index = {}
        self.const_tag = []
        self.const_val = []
        self.theap_tag = []
        self.theap_val = []
        self.columns = {}                  # (root, keys) -> column index
        # what the inline leaf code of the flat / walk2 kernels reads (cbh_check_flat.h flat_leaf): the columns it touches
        # at all, and the ones where an int / uint or a container value sends a lane to the shared evaluator
        self.inline_cols = set()
        self.sensitive_cols = set()
        self.programs = {}                 # dedup key -> entry pc
        self.regex_words = []              # CBH_SEC_REGEX: the DFA tables of constant `matches` patterns, back to back
        self.regex_index = {}              # pattern -> offset of its tables in regex_words
        self.tree_strips = {}              # entry pc of a leaf tree -> (packed ops, n leaves, strip index / 8): _tree_strip
        self._pending_strip = None
        self.dr_names = {}                 # derived role name -> bit
        self.dr_overflow = set()           # derived role names beyond the 64 bits of the mask
        self.unsupported = []              # [(expr text, reason)]
        self.uses_runtime = False
        self.reads_string_bytes = False   # some program may look INSIDE a string (ordering, prefix, size, parsing ...)
        self.req_fields = set()   # cbh_req_field indices some program reads (the host uploads the raw-string fields only then)
        self.has_generic = False           # some program needs the operand-stack interpreter
        self.needs_arena = False           # some program builds a list (filter / map / intersect / except / list +): cbh_blob.h CBH_MF_NEEDS_ARENA
        self.max_stack = 0
        self.max_locals = 0
        # the trace pass (cbh_trace_batch): strings its records refer to - expression texts, variable names, rule FQNs
        self.trace_strings = []
        self.trace_index = {}
        self.trace_unsupported = []        # [(expr text, reason)] of trace programs only
        self.trace_templates = {}          # output word (rule id | not-met << 23) -> (template, number of holes): trace_output_program

    def tid(self, text):
        i = self.trace_index.get(text)
        if i is None:
            i = self.trace_index[text] = len(self.trace_strings)
            self.trace_strings.append(text)
        return i

    # ---- pools ---------------------------------------------------------------------
    def const(self, tag, val):
        k = (tag, val)
        i = self.const_index.get(k)
        if i is None:
            i = len(self.const_tag)
            self.const_index[k] = i
            self.const_tag.append(tag)
            self.const_val.append(val)
        return i

    def zone_table(self, name, table):
        """Offset in the table heap of a zone's [n, from, offset, ...] integers (one copy per zone)."""
        if not hasattr(self, "_zones"):
            self._zones = {}
        at = self._zones.get(name)
        if at is None:
            at = self._zones[name] = len(self.theap_tag)
            vals = [len(table)] + [x for pair in table for x in pair]
            for v in vals:
                self.theap_tag.append(T_INT)
                self.theap_val.append(v & 0xFFFFFFFFFFFFFFFF)
            if at >= 0x40000000:
                raise LoweringError("table heap too large for a zone table")
        return at

    def column(self, root, keys):
        k = (root, tuple(keys))
        i = self.columns.get(k)
        if i is None:
            i = len(self.columns)
            self.columns[k] = i
        return i

    def dr_bit(self, name):
        """Bit of a derived role in the effective-derived-roles mask (64 bits), or None: the table names more than 64 derived
        roles and this one got no bit - what refers to it is lowered so that the requests it concerns are flagged
        (CBH_ST_UNSUPPORTED), the rest of the table still serves."""
        if name not in self.dr_names:
            if len(self.dr_names) >= 64:
                self.dr_overflow.add(name)
                return None
            self.dr_names[name] = len(self.dr_names)
        return self.dr_names[name]

    def _heap_value(self, ast):
        """Constant AST -> (tag, payload) stored in the table heap when it is a container."""
        k = ast[0]
        if k == "lit":
            kind, v = ast[1], ast[2]
            if kind == "null":
                return T_NULL, 0
            if kind == "bool":
                return T_BOOL, int(v)
            if kind == "int":
                return T_INT, v & 0xFFFFFFFFFFFFFFFF
            if kind == "uint":
                return T_UINT, v
            if kind == "double":
                return T_DOUBLE, f64_bits(v)
            if kind == "string":
                return T_STRING, self.sid(v)
            raise _Unsupported("bytes literal")
        if k == "list":
            vals = [self._heap_value(e) for e in ast[1]]
            off = len(self.theap_tag)
            for t, v in vals:
                self.theap_tag.append(t)
                self.theap_val.append(v)
            return T_LIST, cont_payload(HEAP_TABLE, off, len(vals))
        if k == "map":
            ents = []
            for ke, ve in ast[1]:
                kt, kv = self._heap_value(ke)
                if kt != T_STRING:
                    raise _Unsupported("non-string map key literal")
                ents.append(((kt, kv), self._heap_value(ve)))
            off = len(self.theap_tag)
            for (kt, kv), (vt, vv) in ents:
                self.theap_tag.extend((kt, vt))
                self.theap_val.extend((kv, vv))
            return T_MAP, cont_payload(HEAP_TABLE, off, len(ents))
        raise _Unsupported("non-constant container element")

    def _leaf_class(self, op, ka, kb, ci):
        """Shape of a fused leaf for the kernel's straight-line arms (cbh_check_wave.h leaf_fast);
        0 = no special arm.  Operand kinds: 0 constant (index ci), 3 cached column, 4 principal id."""
        eqne = op in (OP_EQ, OP_NE)
        if ka == 3 and kb == 0:
            tag, val = self.const_tag[ci], int(self.const_val[ci])
            if eqne and tag in (T_STRING, T_BOOL):
                return 1
            if tag == T_DOUBLE and op in (OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE):
                return 2
            if op == OP_IN and tag == T_LIST and (val >> 62) == HEAP_TABLE:
                off, n = (val >> 32) & 0x3FFFFFFF, val & 0xFFFFFFFF
                if all(self.theap_tag[off + i] == T_STRING for i in range(n)):
                    return 6 if n <= 3 else 5   # 6: the (at most three) element ids travel in the record itself
            return 0
        if eqne and ka == 3 and kb == 3:
            return 3
        if eqne and ((ka == 3 and kb == 4) or (ka == 4 and kb == 3)):
            return 4
        # membership in a list (or among a map's keys) the REQUEST brings - `"legal-hold" in R.attr.labels`,
        # `R.attr.region in P.attr.regions`: 7 = a string constant, 8 = a column, in a column
        if op == OP_IN and kb == 3 and ka == 0 and self.const_tag[ci] == T_STRING:
            return 7
        if op == OP_IN and kb == 3 and ka in (3, 4):     # (4: the principal's id as the needle)
            return 8
        return 0

    def unsupported_program(self, text, reason):
        """A condition program that marks whoever evaluates it CBH_ST_UNSUPPORTED (the caller's own engine must decide
        that input): for rules the device cannot evaluate faithfully, so that the rest of the table still serves."""
        key = ("__unsupported__", text, reason)
        pc = self.programs.get(key)
        if pc is None:
            self.unsupported.append((text, reason))
            pc = len(self.code)
            self.code.extend([OP_UNSUPPORTED, OP_LEAF, OP_RET])
            self.programs[key] = pc
            self.has_generic = True
            self.max_stack = max(self.max_stack, 1)
        return pc

    def regex(self, pattern):
        """Offset (in u32) of the DFA tables of `pattern` in CBH_SEC_REGEX; compiled once per distinct pattern."""
        off = self.regex_index.get(pattern)
        if off is None:
            words = regex.compile_regex(pattern).words()
            off = self.regex_index[pattern] = len(self.regex_words)
            self.regex_words.extend(words)
        return off

    def _leaf_record(self, w3):
        """[OP_LEAF_BIN word, a0, a1] -> the 8-dword fused-leaf record {w, a0, a1, RET, ctag, clo, chi, class}."""
        words = list(w3) + [OP_RET]
        a = words[0] >> 8
        ka, kb = (a >> 8) & 0xF, (a >> 12) & 0xF
        if (ka == 0) != (kb == 0):
            ci = words[1] if ka == 0 else words[2]
            cv = int(self.const_val[ci]) & 0xFFFFFFFFFFFFFFFF
            cls = self._leaf_class(a & 0xFF, ka, kb, ci)
            if cls in (1, 2, 6):
                self.inline_cols.add(words[1])
                if cls == 2:
                    self.sensitive_cols.add(words[1])
            if cls == 7:
                self.inline_cols.add(words[2])
            if cls == 6:   # column in [<= 3 strings]: the ids instead of the list's heap reference
                off, n = (cv >> 32) & 0x3FFFFFFF, cv & 0xFFFFFFFF
                ids = [int(self.theap_val[off + i]) & 0xFFFFFFFF for i in range(n)] + [0xFFFFFFFF] * (3 - n)
                return words + ids + [6]
            return words + [self.const_tag[ci], cv & 0xFFFFFFFF, cv >> 32, cls]
        cls = self._leaf_class(a & 0xFF, ka, kb, None)
        if cls == 3:
            self.inline_cols.update((words[1], words[2]))
            self.sensitive_cols.update((words[1], words[2]))
        elif cls == 4:
            self.inline_cols.add(words[1] if ka == 3 else words[2])
        elif cls == 8 and ka == 4:
            self.inline_cols.add(words[2])
        elif cls == 8:
            self.inline_cols.update((words[1], words[2]))
            self.sensitive_cols.add(words[1])     # a needle that is not a string (numbers compare across types, containers by content) goes to the evaluator
        return words + [0xFFFFFFFF, 0, 0, cls]

    def _tree_strip(self, pc, words):
        """A condition tree of at most TREE_STRIP_MAX classified leaves (any nesting the op budget holds) also gets, for
        the flat kernel (cbh_check_flat.h flat_tree): its leaves as a strip of consecutive 8-dword fused-leaf records
        behind the program, and its structure as 4-bit ops - 1 = the next leaf of the strip, 2 + k / 5 + k / 8 + k =
        TREE_BEGIN / TREE_ACC / TREE_END of kind k (0 all, 1 any, 2 none), 0 = end - so that a wave evaluates it
        inline and branch-free with no tape reads.  Recorded in tree_strips[pc] = (ops as four u32, n leaves, first
        record's index in 8-dword units); the tape program stays what the general walk runs."""
        ops, leaves, i, depth = [], [], 0, 0
        body = words[:-1]   # without the closing OP_RET
        while i < len(body):
            op, arg = body[i] & 0xFF, body[i] >> 8
            if op == OP_LEAF_BIN:
                leaves.append(self._leaf_record(body[i:i + 3]))
                ops.append(1)
                i += 3
                continue
            if op not in (OP_TREE_BEGIN, OP_TREE_ACC, OP_TREE_END) or arg > 2:
                return
            ops.append({OP_TREE_BEGIN: 2, OP_TREE_ACC: 5, OP_TREE_END: 8}[op] + arg)
            depth += (op == OP_TREE_BEGIN) - (op == OP_TREE_END)
            if depth > 8:
                return
            i += 1
        if not 1 <= len(leaves) <= TREE_STRIP_MAX or len(ops) > 32 or any(rec[7] not in (1, 2, 3, 4, 6, 7, 8) for rec in leaves):
            return
        packed = [0, 0, 0, 0]
        for k, o in enumerate(ops):
            packed[k // 8] |= o << (4 * (k % 8))
        self._pending_strip = (pc, packed, leaves)

    # ---- programs ------------------------------------------------------------------
    def condition_program(self, cond, params: Params, allow_runtime=True):
        """Compile a condition tree; returns the entry pc (deduplicated)."""
        key = (cond, params.key(), allow_runtime)
        pc = self.programs.get(key)
        if pc is not None:
            return pc
        fc = _FuncCompiler(self, params, allow_runtime)
        fc.cond(cond)
        fc.emit(OP_RET)
        pc = len(self.code)
        words = fc.finish(pc)
        is_leaf = len(words) == 4 and (words[0] & 0xFF) == OP_LEAF_BIN
        if is_leaf:
            # A single fused leaf becomes an 8-dword record {op, a0, a1, RET, ctag, clo, chi, class} on an
            # 8-dword boundary: one s_load_dwordx8 fetches the instruction AND the value of its
            # constant operand (ctag = NONE when neither / both operands are constants).
            self.code.extend([OP_RET] * (-len(self.code) % 8))
            pc = len(self.code)
            words = self._leaf_record(words[:3])
        self._pending_strip = None
        if not is_leaf and _is_leaf_tree(words):
            self._tree_strip(pc, words)
        self.code.extend(words)
        if self._pending_strip is not None:
            _, packed, leaves = self._pending_strip
            self.code.extend([OP_RET] * (-len(self.code) % 8))
            self.tree_strips[pc] = (packed, len(leaves), len(self.code) // 8)
            for rec in leaves:
                self.code.extend(rec)
        self._pending_strip = None
        if pc >= COND_PC_MASK:
            raise LoweringError("bytecode tape exceeds 2^30 words")
        if is_leaf:
            pc |= COND_LEAF   # evaluated inline by the table walk (cbh_check_wave.h leaf_fast / eval_cond)
        elif _is_leaf_tree(words):
            pc |= COND_LEAFTREE
        else:
            self.has_generic = True
        self.programs[key] = pc
        self.max_stack = max(self.max_stack, fc.max_depth)
        self.max_locals = max(self.max_locals, fc.max_locals)
        if fc.max_depth > MAX_STACK:
            raise LoweringError("condition needs operand stack depth %d (device limit %d)" % (fc.max_depth, MAX_STACK))
        return pc


    def vars_probe_program(self, params: Params):
        """Did any variable of the params set fail?  The reference evaluates EVERY variable of a rule's policy when the rule
        is visited, read or not (evaluatePrograms, check.go:651-677), and a failure lands in evaluation_errors: the walk asks
        this program (an evaluation site of its own, cbh_check_walk2.h) so that only the inputs it marks need the trace
        pass.  Each definition is evaluated and dropped; OP_LEAF leaves the error status behind.  None = no variables."""
        if not params.ordered_variables:
            return None
        key = ("vars-probe", params.key())
        pc = self.programs.get(key)
        if pc is not None:
            return pc
        fc = _FuncCompiler(self, params, True)
        for _name, text in params.ordered_variables:
            fc.cur_text = text
            d0 = fc.depth
            fc.expr(params.inline(celparser.parse(text)))
            assert fc.depth == d0 + 1
            fc.emit(OP_LEAF)
            fc.emit(OP_POP, 0, -1)
        fc.emit(OP_CONST, self.const(T_BOOL, 1), +1)
        fc.emit(OP_RET)
        pc = len(self.code)
        self.code.extend(fc.finish(pc))
        if pc >= COND_PC_MASK:
            raise LoweringError("bytecode tape exceeds 2^30 words")
        self.has_generic = True
        self.programs[key] = pc
        self.max_stack = max(self.max_stack, fc.max_depth)
        self.max_locals = max(self.max_locals, fc.max_locals)
        if fc.max_depth > MAX_STACK:
            raise LoweringError("variable needs operand stack depth %d (device limit %d)" % (fc.max_depth, MAX_STACK))
        return pc

    # ---- trace programs (cbh_blob.h CBH_SEC_TRACE_*): what the trace pass runs instead of the decision programs.  Same
    # expressions, nothing fused: every leaf ends in OP_LEAF <text id + 1>, which logs a failure under the expression's
    # text; variables keep their OP_VARSCOPE.  They never change what the decision kernels are chosen by or upload
    # (has_generic, uses_runtime, the string / request-field flags): cbh_trace_batch uploads everything.
    def _trace_compile(self, key, build):
        pc = self.programs.get(key)
        if pc is not None:
            return pc
        saved = (self.uses_runtime, self.reads_string_bytes, set(self.req_fields), self.has_generic, self.unsupported)
        self.unsupported = self.trace_unsupported
        try:
            fc = build()
            fc.emit(OP_RET)
            pc = len(self.code)
            self.code.extend(fc.finish(pc))
        finally:
            self.uses_runtime, self.reads_string_bytes, self.req_fields, self.has_generic, self.unsupported = saved
        if pc >= COND_PC_MASK:
            raise LoweringError("bytecode tape exceeds 2^30 words")
        if fc.max_depth > MAX_STACK:
            raise LoweringError("trace program needs operand stack depth %d (device limit %d)" % (fc.max_depth, MAX_STACK))
        self.max_stack = max(self.max_stack, fc.max_depth)
        self.max_locals = max(self.max_locals, fc.max_locals)
        self.programs[key] = pc
        return pc

    def trace_condition_program(self, cond, params: Params, allow_runtime=True):
        assert params.trace

        def build():
            fc = _FuncCompiler(self, params, allow_runtime, trace=True)
            fc.cond(cond)
            return fc
        return self._trace_compile(("trace-cond", cond, params.key(), allow_runtime), build)

    def trace_unsupported_program(self, text, reason):
        def build():
            fc = _FuncCompiler(self, Params(trace=True), True, trace=True)
            fc.cur_text = text
            fc.unsupported(reason)
            fc.emit(OP_LEAF)
            return fc
        return self._trace_compile(("trace-unsupported", text, reason), build)

    def trace_variable_programs(self, params: Params):
        """Entries of one program per variable of the params set, in definition order (evaluatePrograms,
        check.go:651-677): each evaluates the definition and logs a failure under its text."""
        assert params.trace
        out = []
        for name, text in params.ordered_variables:
            def build(text=text):
                fc = _FuncCompiler(self, params, True, trace=True)
                fc.cur_text = text
                fc.value_expr(params.inline(celparser.parse(text)))
                fc.emit(OP_LEAF, self.tid(text) + 1)
                return fc
            out.append(self._trace_compile(("trace-var", text, params.key()), build))
        return out

    def trace_output_program(self, text, params: Params, src, rule_id, not_met):
        """evaluateOutput (check.go:776-807).  What an output expression BUILDS - a list or map literal with computed
        elements, a `"...".format([...])` - is not built on the device: the expression is cut into a template of those
        constructors (kept for the host, trace_templates) and the maximal sub-expressions below them ("holes"), and the
        program evaluates the holes one after the other, logging each value (or error) under the rule's FQN, the id of its
        evaluation key and the hole's number.  An expression without such constructors is one hole."""
        assert params.trace
        ast = params.inline(celparser.parse(text))
        holes = []
        tmpl = _output_template(ast, holes)
        if len(holes) > 64:
            return self.trace_unsupported_program(text, "output expression with more than 64 computed parts")
        if not holes:   # a constant: still one record per visit
            holes, tmpl = [ast], ("hole", 0)
        word = rule_id | ((1 << 23) if not_met else 0)
        self.trace_templates[word] = (tmpl, len(holes))

        def build():
            fc = _FuncCompiler(self, params, True, trace=True)
            fc.cur_text = text
            for j, h in enumerate(holes):
                fc.value_expr(h)
                fc.emit(OP_OUT, self.tid(src))
                fc.word(word)
                fc.word(j)
                if j + 1 < len(holes):
                    fc.emit(OP_POP, 0, -1)
            return fc
        return self._trace_compile(("trace-out", text, params.key(), src, word), build)


def _output_template(ast, holes):
    """-> ("hole", j) | ("const", value) | ("list", [t]) | ("map", [(kt, vt)]) | ("format", fmt, [t]); appends the holes'
    expressions to `holes` in evaluation order."""
    k = ast[0]
    if k == "lit" and ast[1] in ("int", "uint", "double", "string", "bool", "null"):
        return ("const", ast[2] if ast[1] != "null" else None)
    if k == "list" and not _is_const(ast):
        return ("list", [_output_template(e, holes) for e in ast[1]])
    if k == "map" and not _is_const(ast):
        return ("map", [(_output_template(ke, holes), _output_template(ve, holes)) for ke, ve in ast[1]])
    if k == "call" and ast[1] == "format" and ast[2] is not None and ast[2][0] == "lit" and ast[2][1] == "string" \
            and len(ast[3]) == 1 and ast[3][0][0] == "list":
        return ("format", ast[2][2], [_output_template(e, holes) for e in ast[3][0][1]])
    holes.append(ast)
    return ("hole", len(holes) - 1)


class _Unsupported(Exception):
    pass


# OP_HIER kinds (cbh_vm.h hier_pred)
_IP_METHODS = {"family": 1, "isUnspecified": 2, "isLoopback": 3, "isLinkLocalUnicast": 4, "isLinkLocalMulticast": 5, "isGlobalUnicast": 6}
_HIER_PREDICATES = {"ancestorOf": 0, "descendentOf": 1, "immediateParentOf": 2, "immediateChildOf": 3, "siblingOf": 4, "overlaps": 5}

# OP_TS_GETTER kinds (cbh_interp.h): what cel-go's getters return for a timestamp (and, for the last four, a duration)
_TS_GETTERS = {"getFullYear": 0, "getMonth": 1, "getDayOfYear": 2, "getDayOfMonth": 3, "getDate": 4, "getDayOfWeek": 5,
               "getHours": 6, "getMinutes": 7, "getSeconds": 8, "getMilliseconds": 9}


_ATOI = re.compile(r"[+-]?[0-9]+", re.ASCII)


def _fixed_zone_seconds(ast):
    """Seconds east of UTC of a constant time-zone argument cel-go reads without its zone database (timestamp.go
    timeZone(): a string with a ':' is "[+-]hh:mm"), or None.  "UTC" and "" load as UTC."""
    if ast[0] != "lit" or ast[1] != "string":
        return None
    tz = ast[2]
    if tz in ("UTC", ""):
        return 0
    if ":" not in tz:
        return None
    hh, _, mm = tz.partition(":")
    # strconv.Atoi's syntax, not Python's int(): ASCII digits behind an optional sign - no blanks, no '_', no other
    # scripts' digits (cel-go answers those with an error, i.e. a condition that is false); anything else stays UNSUPPORTED
    if not (_ATOI.fullmatch(hh) and _ATOI.fullmatch(mm)):
        return None
    h, m = int(hh), int(mm)
    if not (-23 <= h <= 23 and 0 <= m <= 59):
        return None
    neg = tz[0] == "-"     # cel-go tests val[0]
    return (-1 if neg else 1) * (abs(h) * 3600 + m * 60)


_ZONE_FROM, _ZONE_TO = -2208988800, 4102444800   # 1900-01-01 .. 2100-01-01 UTC: what a lowered zone table covers
_zone_tables = {}


def _named_zone_table(name):
    """The UTC offsets of an IANA zone as [(from unix second, seconds east of UTC), ...] over 1900 .. 2100 - what Go's
    time.LoadLocation + Time.In amount to for the getters - read off the zone database through zoneinfo: the offset is
    sampled day by day and every change bisected to the second.  None when the name is not in the database."""
    if name in _zone_tables:
        return _zone_tables[name]
    import datetime
    import zoneinfo
    table = None
    try:
        if name and "/" != name[:1] and ".." not in name and name != "Local":
            zi = zoneinfo.ZoneInfo(name)
            utc = datetime.timezone.utc

            def off(t):
                return int(datetime.datetime.fromtimestamp(t, utc).astimezone(zi).utcoffset().total_seconds())
            table = [(_ZONE_FROM, off(_ZONE_FROM))]
            t, cur = _ZONE_FROM, table[0][1]
            while t < _ZONE_TO:
                nt = min(t + 86400, _ZONE_TO)
                o = off(nt)
                while o != cur:   # a change in (t, nt]: bisect to its second (twice when two changes share the day)
                    lo, hi = t, nt
                    while hi - lo > 1:
                        mid = (lo + hi) // 2
                        if off(mid) == cur:
                            lo = mid
                        else:
                            hi = mid
                    cur = off(hi)
                    table.append((hi, cur))
                    t = hi
                t = nt
    except Exception:   # unknown zone, no zone database
        table = None
    _zone_tables[name] = table
    return table


def _int_lit_as_double(ast):
    if ast[0] == "lit" and ast[1] in ("int", "uint") and abs(ast[2]) <= (1 << 53):
        return ("lit", "double", float(ast[2]))
    return ast


def _builds_string(ast):
    """Is the expression visibly a string: a literal, lowerAscii / upperAscii of something, or a concatenation with one."""
    k = ast[0]
    if k == "lit":
        return ast[1] == "string"
    if k == "call":
        if ast[1] in ("lowerAscii", "upperAscii", "trim") and ast[2] is not None and not ast[3]:
            return True
        return ast[2] is not None and ((ast[1] in ("charAt",) and len(ast[3]) == 1) or (ast[1] == "substring" and len(ast[3]) in (1, 2))
                                       or (ast[1] == "replace" and len(ast[3]) == 2))
    if k == "bin" and ast[1] == "+":
        return _builds_string(ast[2]) or _builds_string(ast[3])
    return False


def _builds_list(ast):
    """Is the expression visibly a list: a literal, filter / map, intersect / except, or a concatenation of such."""
    k = ast[0]
    if k == "list":
        return True
    if k == "comp":
        return ast[1] in ("filter", "map", "transformList")
    if k == "call":
        return ast[1] in ("intersect", "except", "slice") or (ast[1] == "reverse" and ast[2] is not None and _builds_list(ast[2])) \
            or (ast[1] == "range" and ast[2] is not None and ast[2][0] == "ident" and ast[2][1] == "lists")
    if k == "bin" and ast[1] == "+":
        return _builds_list(ast[2]) or _builds_list(ast[3])
    return False


def _is_const(ast):
    k = ast[0]
    if k == "lit":
        return ast[1] != "bytes"
    if k == "list":
        return all(_is_const(e) for e in ast[1])
    if k == "map":
        return all(ke[0] == "lit" and ke[1] == "string" and _is_const(ve) for ke, ve in ast[1])
    return False


class _FuncCompiler:
    def __init__(self, pb: ProgramBuilder, params: Params, allow_runtime: bool, trace=False):
        self.pb = pb
        self.params = params
        self.allow_runtime = allow_runtime
        self.trace = trace
        self.out = []          # list of ints or ('label', id) / ('ref', op, id)
        self.depth = 0
        self.max_depth = 0
        self.labels = {}
        self.nlabels = 0
        self.locals = {}       # var name -> slot
        self.max_locals = 0
        self.iter_depth = 0
        self.tree_depth = 0
        self.cur_text = ""

    # -- emission helpers
    def emit(self, op, arg=0, delta=0):
        if op not in _ID_ONLY_OPS:   # conservatively: everything that is not known to work on ids / numbers alone
            self.pb.reads_string_bytes = True
        self.out.append(op | (arg << 8))
        self.depth += delta
        self.max_depth = max(self.max_depth, self.depth)

    def word(self, w):
        self.out.append(w)

    def new_label(self):
        self.nlabels += 1
        return self.nlabels

    def place(self, lab):
        self.out.append(("label", lab))

    def emit_ref(self, op, lab, delta=0):
        self.out.append(("ref", op, lab))
        self.depth += delta
        self.max_depth = max(self.max_depth, self.depth)

    def word_ref(self, lab):
        """A full word holding the address of ``lab``."""
        self.out.append(("wref", lab, None))

    def word_ref_packed(self, lab, low):
        """A word holding ``low | address(lab) << 8``."""
        self.out.append(("wref", lab, low))

    def finish(self, base):
        pos, addr = 0, {}
        for x in self.out:
            if isinstance(x, tuple) and x[0] == "label":
                addr[x[1]] = base + pos
            else:
                pos += 1
        code = []
        for x in self.out:
            if isinstance(x, tuple):
                if x[0] == "label":
                    continue
                if x[0] == "ref":
                    code.append(x[1] | (addr[x[2]] << 8))
                elif x[0] == "wrefhi":
                    code.append(addr[x[1]] | (x[2] << 30))
                elif x[2] is None:
                    code.append(addr[x[1]])
                else:
                    code.append(x[2] | (addr[x[1]] << 8))
            else:
                code.append(x)
        return code

    def unsupported(self, reason):
        self.pb.unsupported.append((self.cur_text, reason))
        self.emit(OP_UNSUPPORTED, 0, +1)

    # -- condition trees (check.go:679-756)
    def cond(self, c):
        if c is None:
            self.emit(OP_CONST, self.pb.const(T_BOOL, 1), +1)
            return
        op = c[0]
        if op == "expr":
            self.cur_text = c[1]
            ast = self.params.inline(celparser.parse(c[1]))
            if not self.trace and self._fused_leaf(ast):
                return
            d0 = self.depth
            self.expr(ast)
            assert self.depth == d0 + 1, (c[1], self.depth, d0)
            self.emit(OP_LEAF, (self.pb.tid(c[1]) + 1) if self.trace else 0)
            return
        kids = c[1]
        if not kids:
            self.emit(OP_CONST, self.pb.const(T_BOOL, 1 if op in ("all", "none") else 0), +1)
            return
        # control flow is wave-uniform: no jumps; the interpreter keeps a per-lane "tree-live"
        # predicate so that children after the deciding one record no errors (check.go:697-749)
        kind = TREE_KINDS[op]
        self.tree_depth += 1
        if self.tree_depth > 16:
            raise LoweringError("condition tree nests deeper than 16 levels")
        self.emit(OP_TREE_BEGIN, kind)
        for kid in kids:
            self.cond(kid)
            self.emit(OP_TREE_ACC, kind, -1)
        self.emit(OP_TREE_END, kind, +1)
        self.tree_depth -= 1

    # -- fused leaves: `operand <cmp|in> operand` with operands read straight from a column, a
    #    request string field or the constant pool (the shape of almost every real condition)
    def _simple_operand(self, ast):
        """-> (kind, arg) or None.  Kinds: 0 constant, 1 column, 2 request string field, and the two
        the decision kernel serves without a memory access (cbh_check_wave.h leaf_fast): 3 = a column
        of the kernel's LDS column cache (the first CACHE_COLS columns), 4 = the principal id."""
        k = ast[0]
        if k == "lit" or (k in ("list", "map") and _is_const(ast)):
            try:
                t, v = self.pb._heap_value(ast)
            except _Unsupported:
                return None
            return 0, self.pb.const(t, v)
        if k in ("select", "index"):
            p = self._path(ast)
            if p is not None and p[0] == "col":
                ci = self.pb.column(p[1], p[2])
                return (3 if ci < CACHE_COLS else 1), ci
            if p is not None and p[0] == "req":
                return (4, 0) if p[1] == RQ_PRINCIPAL_ID else (2, p[1])
        return None

    def _fused_leaf(self, ast) -> bool:
        if ast[0] != "bin" or ast[1] not in ("==", "!=", "<", "<=", ">", ">=", "in"):
            return False
        if self.locals:
            return False
        lhs, rhs = ast[2], ast[3]
        if ast[1] != "in":
            # request attributes always arrive as doubles (structpb); an int literal that a double
            # represents exactly compares identically as a double under CEL's cross-type numeric
            # comparison, and lets the device take the same-type fast path
            lhs, rhs = _int_lit_as_double(lhs), _int_lit_as_double(rhs)
        a = self._simple_operand(lhs)
        b = self._simple_operand(rhs)
        if a is None or b is None:
            return False
        if ast[1] in ("<", "<=", ">", ">="):
            # an ordering compares string CONTENTS unless one side is a constant that is not a string
            tags = [self.pb.const_tag[x[1]] for x in (a, b) if x[0] == 0]
            if not tags or any(t == T_STRING for t in tags):
                self.pb.reads_string_bytes = True
        self.emit(OP_LEAF_BIN, _BINOPS[ast[1]] | (a[0] << 8) | (b[0] << 12), +1)
        self.word(a[1])
        self.word(b[1])
        return True

    def value_expr(self, ast):
        """An expression whose VALUE is wanted (a part of an output expression, a variable's definition - trace programs).  As
        expr(), except that `runtime.effectiveDerivedRoles` may be the whole of it: the device pushes the derived-role mask of the
        scope being walked under a tag of its own (OP_EDRVAL, CBH_T_EDRSET) and the host spells the sorted names (check.go:593-610
        builds the list from the set the same way); inside a larger expression the list stays outside the subset, as before."""
        if self.trace and self.allow_runtime:
            if ast[0] in ("select", "index") and self._path(ast) == ("edr",):
                self.pb.uses_runtime = True
                return self.emit(OP_EDRVAL, 0, +1)
            if ast[0] == "varscope":   # a variable that IS the list (V.derivedRoles: runtime.effectiveDerivedRoles), read as a whole
                self.value_expr(ast[2])
                return self.emit(OP_VARSCOPE, self.pb.tid(ast[1]) | (ast[3] << 23))
        return self.expr(ast)

    # -- paths
    def _path(self, ast):
        """Resolve a select/index chain to ('col', root, keys) | ('req', field) | ('roles',) |
        ('edr',) | None."""
        keys = []
        n = ast
        while True:
            if n[0] == "select":
                keys.append(n[2])
                n = n[1]
            elif n[0] == "index" and n[2][0] == "lit" and n[2][1] == "string":
                keys.append(n[2][2])
                n = n[1]
            else:
                break
        if n[0] != "ident" or n[1] in self.locals:
            return None
        keys.reverse()
        base = n[1]
        if base == "request":
            if not keys:
                return None
            top = keys[0]
            if top == "principal":
                base, keys = "P", keys[1:]
            elif top == "resource":
                base, keys = "R", keys[1:]
            elif top in ("aux_data", "auxData"):
                if len(keys) >= 2 and keys[1] == "jwt":
                    return ("col", "J", tuple(keys[2:]))
                if len(keys) >= 2 and keys[1] == "jwts":     # AuxData.jwts: name -> {claims: map} (engine.proto:314-321)
                    return ("col", "S", tuple(keys[2:]))
                return None
            else:
                return None
        if base in ("P", "R"):
            if not keys:
                return None
            f = keys[0]
            if f == "attr":
                return ("col", base, tuple(keys[1:]))
            if len(keys) == 1:
                if base == "P" and f == "roles":
                    return ("roles",)
                fields = _P_FIELDS if base == "P" else _R_FIELDS
                if f in fields:
                    self.pb.req_fields.add(fields[f])
                    return ("req", fields[f])
            return None
        if base == "runtime" and len(keys) == 1 and keys[0] in ("effectiveDerivedRoles", "effective_derived_roles"):
            return ("edr",)
        if base in ("G", "globals") and self.pb.per_call_globals:
            return ("col", "G", tuple(keys))
        return None

    def _stringy(self, ast):
        """Visibly a string: a literal / case mapping / concatenation with one (_builds_string), or one of the request's string
        fields (P.id, R.id, R.kind, scopes, policy versions)."""
        if _builds_string(ast):
            return True
        if ast[0] == "bin" and ast[1] == "+":
            return self._stringy(ast[2]) or self._stringy(ast[3])
        if ast[0] in ("select", "index"):
            p = self._path(ast)
            return p is not None and p[0] == "req"
        return False

    # -- expressions
    def expr(self, ast):  # noqa: C901
        try:
            self._expr(ast)
        except _Unsupported as e:
            raise LoweringError("internal: unsupported escaped: %s" % e)

    def _expr(self, ast):  # noqa: C901
        k = ast[0]
        pb = self.pb
        if k == "lit" or ((k == "list" or k == "map") and _is_const(ast)):
            try:
                t, v = pb._heap_value(ast)
            except _Unsupported as e:
                return self.unsupported(str(e))
            return self.emit(OP_CONST, pb.const(t, v), +1)
        if k in ("list", "map"):
            return self.unsupported("container literal with non-constant elements")
        if k == "ident":
            if ast[1] in self.locals:
                return self.emit(OP_LOCAL, self.locals[ast[1]], +1)
            return self.unsupported("identifier %s used as a value" % ast[1])
        if k in ("select", "index"):
            p = self._path(ast)
            if p is not None:
                if p[0] == "col":
                    return self.emit(OP_COL, pb.column(p[1], p[2]), +1)
                if p[0] == "req":
                    return self.emit(OP_REQSTR, p[1], +1)
                if p[0] == "roles":
                    return self.emit(OP_ROLES, 0, +1)
                return self.unsupported("runtime.effectiveDerivedRoles used other than `name in ...`")
            if k == "select":
                self._expr(ast[1])
                return self.emit(OP_SELECT, pb.sid(ast[2]))
            self._expr(ast[1])
            self._expr(ast[2])
            return self.emit(OP_INDEX, 0, -1)
        if k == "has":
            p = self._path(("select", ast[1], ast[2]))
            if p is not None and p[0] == "col" and len(p[2]) >= 1:
                return self.emit(OP_HASCOL, pb.column(p[1], p[2]), +1)
            if p is not None:
                return self.unsupported("has() on a request field")
            self._expr(ast[1])
            return self.emit(OP_HASSEL, pb.sid(ast[2]))
        if k == "not":
            self._expr(ast[1])
            return self.emit(OP_NOT)
        if k == "neg":
            self._expr(ast[1])
            return self.emit(OP_NEG)
        if k in ("and", "or"):
            # both sides are evaluated (uniform control flow); AND/OR absorb errors like CEL
            self._expr(ast[1])
            self._expr(ast[2])
            return self.emit(OP_AND if k == "and" else OP_OR, 0, -1)
        if k == "tern":
            self._expr(ast[1])
            self._expr(ast[2])
            self._expr(ast[3])
            return self.emit(OP_TERN, 0, -2)
        if k == "bin":
            op = ast[1]
            if op == "in":
                p = self._path(ast[3]) if ast[3][0] in ("select", "index") else None
                if p is not None and p[0] == "edr":
                    if ast[2][0] == "lit" and ast[2][1] == "string" and self.allow_runtime:
                        pb.uses_runtime = True
                        bit = pb.dr_bit(ast[2][2])
                        if bit is None:
                            return self.unsupported("runtime.effectiveDerivedRoles names a derived role beyond the 64 the device mask holds")
                        return self.emit(OP_EDRHAS, bit, +1)
                    return self.unsupported("runtime.effectiveDerivedRoles membership with a non-constant name")
            if op in ("==", "!=") and self.allow_runtime:
                # runtime.effectiveDerivedRoles == [constant names] (either order): the runtime list is the SORTED names of the
                # scope's effective derived roles (check.go:601-607), so the comparison is one mask compare
                for lhs, rhs in ((ast[2], ast[3]), (ast[3], ast[2])):
                    p = self._path(lhs) if lhs[0] in ("select", "index") else None
                    if p is not None and p[0] == "edr" and rhs[0] == "list" and all(e[0] == "lit" and e[1] == "string" for e in rhs[1]):
                        names = [e[2] for e in rhs[1]]
                        never = any(a >= b for a, b in zip(names, names[1:]))   # not strictly ascending: never equal
                        mask = 0
                        bits = [pb.dr_bit(nm) for nm in names]
                        if any(b is None for b in bits):
                            return self.unsupported("runtime.effectiveDerivedRoles compared with a derived role beyond the 64 the device mask holds")
                        for b in bits:
                            mask |= 1 << b
                        pb.uses_runtime = True
                        self.emit(OP_EDREQ, pb.const(T_UINT, mask) | (0x80000000 if never else 0), +1)
                        return self.emit(OP_NOT) if op == "!=" else None
            if op in ("==", "!="):
                # `a.lowerAscii() == b`, `a == b.upperAscii()` ...: the device builds no strings, it compares the bytes
                # through the case mapping (cel-go ext/strings.go lowerAscii / upperAscii: ASCII letters only)
                modes, sides = [], []
                for side in (ast[2], ast[3]):
                    m = 0
                    if side[0] == "call" and side[1] in ("lowerAscii", "upperAscii") and side[2] is not None and not side[3] \
                            and not _builds_string(side[2]) \
                            and not (side[2][0] == "ident" and side[2][1] in ("strings",) and side[2][1] not in self.locals):
                        m, side = (1 if side[1] == "lowerAscii" else 2), side[2]
                    modes.append(m); sides.append(side)
                if modes[0] or modes[1]:
                    self._expr(sides[0])
                    self._expr(sides[1])
                    return self.emit(OP_STREQ_CASE, modes[0] | (modes[1] << 2) | ((1 if op == "!=" else 0) << 4), -1)
            if op in ("==", "!="):
                # hierarchy(a).commonAncestors(hierarchy(b)) == hierarchy(c) (either order): the ancestors are never built - the
                # device walks the segments the two strings share and compares them with c's (cbh_vm.h hier_common_eq)
                def hier_of(x):
                    return x[3][0] if x[0] == "call" and x[1] == "hierarchy" and x[2] is None and len(x[3]) == 1 else None
                for lhs, rhs in ((ast[2], ast[3]), (ast[3], ast[2])):
                    if lhs[0] == "call" and lhs[1] == "commonAncestors" and lhs[2] is not None and len(lhs[3]) == 1:
                        a, b, c3 = hier_of(lhs[2]), hier_of(lhs[3][0]), hier_of(rhs)
                        if a is not None and b is not None and c3 is not None:
                            self.pb.reads_string_bytes = True
                            self._expr(a)
                            self._expr(b)
                            self._expr(c3)
                            self.emit(OP_HIERCOMMON, 0, -2)
                            return self.emit(OP_NOT) if op == "!=" else None
            if op == "+" and (self._stringy(ast[2]) or self._stringy(ast[3])):
                # string concatenation where one side is visibly a string: a rope (cbh_vm.h) - the parts side by side in the
                # lane's arena, no byte copied; equality, `in`, startsWith / endsWith / contains and size() read ropes
                self.pb.needs_arena = True
                self.pb.reads_string_bytes = True
                self._expr(ast[2])
                self._expr(ast[3])
                return self.emit(OP_STRCAT, 0, -1)
            if op == "+" and (_builds_list(ast[2]) or _builds_list(ast[3])):
                # list concatenation where one side is visibly a list: built in the lane's arena (any other `+` stays arithmetic,
                # which flags lists it meets at run time)
                self.pb.needs_arena = True
                self._expr(ast[2])
                self._expr(ast[3])
                return self.emit(OP_LISTOP, 2, -1)
            self._expr(ast[2])
            self._expr(ast[3])
            return self.emit(_BINOPS[op], 0, -1)
        if k == "call":
            return self._call(ast)
        if k == "comp":
            return self._comp(ast)
        if k == "varscope":   # trace programs: the inlined definition of variable ast[1]
            self._expr(ast[2])
            return self.emit(OP_VARSCOPE, pb.tid(ast[1]) | (ast[3] << 23))
        return self.unsupported("%s expression" % k)

    def _call(self, ast):
        _, name, target, args = ast
        if name == "__unsupported__":
            return self.unsupported("reference to an undefined variable or constant")
        if name == "__undef__":   # trace programs: an undefined global reads "undefined field '<name>'"
            self.emit(OP_CONST, self.pb.const(T_NULL, 0), +1)
            self.emit(OP_NEG)
            return self.emit(OP_VARSCOPE, self.pb.tid(args[0][2]))
        if name == "__error__":
            # evaluates to a CEL error without marking the tuple unsupported
            self.emit(OP_CONST, self.pb.const(T_NULL, 0), +1)
            return self.emit(OP_NEG)
        allargs = ([target] if target is not None else []) + list(args)
        n = len(allargs)

        def unary(op):
            self._expr(allargs[0])
            self.emit(op)

        def binary(op):
            self._expr(allargs[0])
            self._expr(allargs[1])
            self.emit(op, 0, -1)

        # cel-go ext.Network on strings the request supplies (constant operands were folded by the lowering, cel/fold.py)
        def ip_of(x):
            return x[3][0] if x[0] == "call" and x[1] == "ip" and x[2] is None and len(x[3]) == 1 else None
        if name == "isIP" and target is None and n in (1, 2):
            fam = 0
            if n == 2:
                if args[1][0] != "lit" or args[1][1] != "int" or args[1][2] not in (4, 6):
                    return self.unsupported("function isIP/2 with a computed version")
                fam = 8 if args[1][2] == 4 else 9
            self.pb.reads_string_bytes = True
            self._expr(args[0])
            return self.emit(OP_IPFN, fam)
        if name == "isCanonical" and target == ("ident", "ip") and "ip" not in self.locals and n == 2:
            self.pb.reads_string_bytes = True
            self._expr(args[0])
            return self.emit(OP_IPFN, 7)
        if name in _IP_METHODS and target is not None and ip_of(target) is not None and n == 1:
            self.pb.reads_string_bytes = True
            self._expr(ip_of(target))
            return self.emit(OP_IPFN, _IP_METHODS[name])
        if name == "containsIP" and target is not None and n == 2 and target[0] == "call" and target[1] == "cidr" and target[2] is None \
                and len(target[3]) == 1:
            self.pb.reads_string_bytes = True
            self._expr(target[3][0])
            self._expr(ip_of(args[0]) if ip_of(args[0]) is not None else args[0])
            return self.emit(OP_IPFN, 10, -1)
        ns = target is not None and target[0] == "ident" and target[1] in ("sets", "math", "lists", "base64", "strings", "regex") \
            and target[1] not in self.locals
        if not ns:
            if target is not None and ((name == "substring" and n in (2, 3)) or (name == "charAt" and n == 2) or (name == "trim" and n == 1)):
                # a window of the string, as a rope (cbh_vm.h): indices count code points (cel-go ext/strings.go)
                self.pb.needs_arena = True
                self.pb.reads_string_bytes = True
                for x in allargs:
                    self._expr(x)
                return self.emit(OP_STRVIEW, {"substring": n - 2, "charAt": 2, "trim": 3}[name], -(n - 1))
            if target is not None and name == "replace" and n == 3:
                self.pb.needs_arena = True
                self.pb.reads_string_bytes = True
                for x in allargs:
                    self._expr(x)
                return self.emit(OP_STRREPLACE, 0, -2)
            if name == "size" and n == 1:
                return unary(OP_SIZE)
            if name in ("startsWith", "endsWith", "contains") and n == 2:
                return binary({"startsWith": OP_STARTSWITH, "endsWith": OP_ENDSWITH, "contains": OP_CONTAINS}[name])
            if name == "timestamp" and n == 1 and target is None:
                return unary(OP_TIMESTAMP)
            if name == "duration" and n == 1 and target is None:
                return unary(OP_DURATION)
            if name == "timeSince" and n == 1:
                return unary(OP_TIMESINCE)
            if name == "now" and n == 0:
                return self.emit(OP_NOW, 0, +1)
            if name in ("indexOf", "lastIndexOf") and n == 2 and target is not None:
                # cel-go ext/strings.go: the code-point index of the first / last occurrence, -1 without one
                self._expr(allargs[0])
                self._expr(allargs[1])
                return self.emit(OP_INDEXOF, 0 if name == "indexOf" else 1, -1)
            if name == "matches" and n == 2:
                # cel-go `matches` = RE2 MatchString (an unanchored search).  A pattern that is a constant of the policy
                # is compiled to a byte-level DFA here (regex.py); the device walks one table lookup per byte.
                pat = allargs[1]
                if pat[0] != "lit" or pat[1] != "string":
                    return self.unsupported("function matches/2 with a computed pattern")
                try:
                    off = self.pb.regex(pat[2])
                except (regex.Unsupported, regex.Invalid) as e:
                    return self.unsupported("function matches/2: %s" % e)
                self._expr(allargs[0])
                self.emit(OP_MATCHES)
                self.word(off)
                return None
            if name in _HIER_PREDICATES and n == 2 and all(
                    x[0] == "call" and x[1] == "hierarchy" and x[2] is None and len(x[3]) == 1 for x in (target, args[0])):
                # hierarchy(a).ancestorOf(hierarchy(b)) and its siblings (internal/conditions/types/hierarchy.go:259-385)
                # over the two dot-delimited STRINGS: the predicates only compare segment prefixes, no list is built.
                # hierarchy(list), a custom delimiter, indexing, size and commonAncestors stay outside the subset.
                self._expr(target[3][0])
                self._expr(args[0][3][0])
                return self.emit(OP_HIER, _HIER_PREDICATES[name], -1)
            if name in _TS_GETTERS and target is not None and n in (1, 2):
                # timestamp / duration getters (cel-go timestamp.go / duration.go).  The time zone argument is resolved
                # here: absent or "UTC" = no offset, "+hh:mm" / "-hh:mm" = a fixed offset; an IANA name needs the
                # zone database and a computed one cannot be resolved at lowering time - both outside the subset.
                off = 0
                if n == 2:
                    off = _fixed_zone_seconds(args[0])
                    if off is None and args[0][0] == "lit" and args[0][1] == "string" and ":" not in args[0][2]:
                        # an IANA name: its offsets over 1900 .. 2100 travel in the table's constant heap
                        # ([n, from_0, offset_0, from_1, offset_1, ...]); the device looks the timestamp up (cbh_interp.h)
                        table = _named_zone_table(args[0][2])
                        if table is None:
                            return self.unsupported("function %s/%d: time zone %r is not in the zone database" % (name, n, args[0][2]))
                        self._expr(target)
                        self.emit(OP_TS_GETTER, _TS_GETTERS[name])
                        self.word(0x40000000 | self.pb.zone_table(args[0][2], table))
                        return None
                    if off is None:
                        return self.unsupported("function %s/%d with a computed time zone" % (name, n))
                self._expr(target)
                self.emit(OP_TS_GETTER, _TS_GETTERS[name])
                self.word(off & 0x3FFFFFFF if off >= 0 else off & 0xFFFFFFFF)
                return None
            if name == "int" and n == 1 and target is None:
                return unary(OP_TOINT)
            if name == "uint" and n == 1 and target is None:   # (the same opcode, argument 1)
                self._expr(allargs[0])
                return self.emit(OP_TOINT, 1)
            if name == "double" and n == 1 and target is None:
                return unary(OP_TODOUBLE)
            if name == "dyn" and n == 1 and target is None:
                return self._expr(allargs[0])
            if name == "inIPAddrRange" and n == 2:
                return binary(OP_INIPRANGE)
            if name in ("hasIntersection", "has_intersection") and n == 2:
                return binary(OP_HASINTERSECTION)
            if name in ("lowerAscii", "upperAscii") and n == 1 and target is not None:
                # as a value (not under == / !=, which compares through the mapping directly): a rope that reads the string
                # through the mapping (cel-go ext/strings.go: ASCII letters only)
                self.pb.needs_arena = True
                self.pb.reads_string_bytes = True
                self._expr(target)
                return self.emit(OP_STRCASE, 1 if name == "lowerAscii" else 2)
            if name == "reverse" and n == 1 and _builds_list(allargs[0]):   # (strings reverse too: only where the operand is visibly a list)
                self.pb.needs_arena = True
                self._expr(allargs[0])
                return self.emit(OP_LISTFN, 0)
            if name == "slice" and n == 3:
                for x in allargs:
                    self._expr(x)
                return self.emit(OP_LISTFN, 1, -2)
            if name in ("intersect", "except") and n == 2:   # cerbos_lib.go: elements of the first list (not) in the second, in order
                self.pb.needs_arena = True
                self._expr(allargs[0])
                self._expr(allargs[1])
                return self.emit(OP_LISTOP, 0 if name == "intersect" else 1, -1)
            if name in ("isSubset", "is_subset") and n == 2:
                return binary(OP_ISSUBSET)
        elif target[1] == "lists" and name == "range" and n == 2:   # lists.range(n) = [0 .. n)
            self.pb.needs_arena = True
            self._expr(args[0])
            return self.emit(OP_LISTFN, 2)
        elif target[1] == "sets" and n == 3:
            a, b = args
            if name == "intersects":
                self._expr(a); self._expr(b)
                return self.emit(OP_HASINTERSECTION, 0, -1)
            if name == "contains":   # sets.contains(a, b): every element of b is in a
                self._expr(b); self._expr(a)
                return self.emit(OP_ISSUBSET, 0, -1)
            if name == "equivalent":   # sets.equivalent(a, b) = contains(a, b) && contains(b, a) (cel-go ext/sets.go)
                self._expr(b); self._expr(a)
                self.emit(OP_ISSUBSET, 0, -1)
                self._expr(a); self._expr(b)
                self.emit(OP_ISSUBSET, 0, -1)
                return self.emit(OP_AND, 0, -1)
        return self.unsupported("function %s/%d" % (name, n))

    def _comp(self, ast):
        _, kind, target, vars_, args = ast
        kinds = {"all": IT_ALL, "exists": IT_EXISTS, "exists_one": IT_EXISTS_ONE, "existsOne": IT_EXISTS_ONE,
                 "filter": IT_FILTER, "map": IT_MAP, "transformList": IT_MAP}
        if kind not in kinds or (kind == "filter" and len(args) != 1) or len(args) not in (1, 2):
            return self.unsupported("macro %s" % kind)   # (transformMap / transformMapEntry / sortBy build maps or order: not on the device)
        kind_id = kinds[kind]
        if kind in ("filter", "map", "transformList"):
            self.pb.needs_arena = True   # the result list is built in the lane's arena (cbh_vm.h)
            if len(args) == 2:           # map(x, pred, expr) / transformList(i, v, pred, expr)
                kind_id = IT_MAP_FILTER
        elif len(args) != 1:
            return self.unsupported("macro %s" % kind)
        if self.iter_depth >= MAX_ITERS or len(self.locals) + len(vars_) > MAX_LOCALS:
            return self.unsupported("comprehension nesting beyond the device limits")
        slot = self.iter_depth
        self._expr(target)
        loop, end = self.new_label(), self.new_label()
        self.emit(OP_ITER_BEGIN, slot, -1)
        self.word(kind_id)
        saved = dict(self.locals)
        slots = []
        for v in vars_:
            s = len(self.locals)
            self.locals[v] = s
            slots.append(s)
        self.max_locals = max(self.max_locals, len(self.locals))
        self.iter_depth += 1
        self.place(loop)
        self.emit(OP_ITER_NEXT, slot)
        self.word_ref(end)
        self.word(slots[0] | ((slots[1] if len(slots) > 1 else 0) << 8) | (len(slots) << 16))
        d0 = self.depth
        for body in args:   # the predicate (all / exists / filter), the element (map), or predicate then element
            self._expr(body)
        assert self.depth == d0 + len(args)
        self.emit(OP_ITER_ACC, slot, -len(args))
        self.out.append(("wrefhi", loop, slots[0]))   # loop address | the loop variable's local << 30 (filter keeps its value)
        self.place(end)
        self.emit(OP_ITER_END, slot, +1)
        self.iter_depth -= 1
        self.locals = saved
