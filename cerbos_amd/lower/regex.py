"""RE2 pattern -> byte-level DFA tables for the device's ``matches`` (cel-go ``matches`` = RE2 ``MatchString``: an
unanchored search).

The reference compiles patterns with Go's regexp (RE2 syntax, github.com/google/cel-go ext / stdlib ``matches``); the
device cannot run a backtracking or NFA matcher per lane cheaply, so a pattern that is a *constant* of the policy is
compiled here, at lowering time, into a deterministic automaton over UTF-8 bytes:

    parse (RE2 subset)  ->  Thompson NFA over byte sets, with ^ / $ as assertions  ->  subset construction with the
    start state re-injected at every position (search semantics)  ->  byte equivalence classes  ->  u32 tables

The device walks one table lookup per input byte (cbh_vm.h regex_match).  Everything outside the subset raises
``Unsupported`` and the expression is flagged UNSUPPORTED - never answered approximately:
case folding and other flags, ``\\b`` / ``\\B``, Unicode classes (``\\p{..}``), POSIX classes, non-ASCII characters inside
a bracket class, ``\\Q..\\E``, octal escapes, automata beyond MAX_STATES.

Subset semantics worth spelling out (RE2, not PCRE / Python): ``.`` and negated classes match one whole code point
(never ``\\n`` for ``.``), ``\\d \\w \\s`` are ASCII-only with ``\\s`` = ``[\\t\\n\\f\\r ]``, ``$`` matches only at the very
end of the text, a ``{`` that does not open a valid repetition is a literal, repetition counts are capped at 1000.
"""
from __future__ import annotations

MAX_STATES = 1024          # DFA states per pattern
MAX_NFA = 20000            # NFA states (counted repetitions are expanded)
MAX_REPEAT = 1000          # RE2's own cap


class Unsupported(Exception):
    """Valid RE2 the device subset does not cover."""


class Invalid(Exception):
    """Not a valid RE2 pattern (the reference rejects the policy at compile time)."""


_DIGIT = frozenset(range(0x30, 0x3A))
_WORD = frozenset(list(range(0x30, 0x3A)) + list(range(0x41, 0x5B)) + list(range(0x61, 0x7B)) + [0x5F])
_SPACE = frozenset(b"\t\n\f\r ")
_ASCII = frozenset(range(0x80))
_ESC_LIT = {"n": 0x0A, "t": 0x09, "r": 0x0D, "f": 0x0C, "v": 0x0B, "a": 0x07}


class _Set:
    """A set of code points: ASCII members + "every non-ASCII code point" (negated classes, ``.``)."""
    __slots__ = ("ascii", "non_ascii")

    def __init__(self, ascii_=frozenset(), non_ascii=False):
        self.ascii, self.non_ascii = frozenset(ascii_), non_ascii

    def negate(self):
        return _Set(_ASCII - self.ascii, not self.non_ascii)

    def union(self, o):
        return _Set(self.ascii | o.ascii, self.non_ascii or o.non_ascii)


# ---- parser: pattern -> AST ------------------------------------------------------------------------------------
# ("set", _Set) | ("bytes", b"...") | ("cat", [..]) | ("alt", [..]) | ("rep", node, min, max|None) | ("bol",) | ("eol",) | ("empty",)
class _Parser:
    def __init__(self, pattern: str):
        self.s, self.i = pattern, 0

    def peek(self):
        return self.s[self.i] if self.i < len(self.s) else None

    def parse(self):
        node = self.alt()
        if self.i != len(self.s):
            raise Invalid("unexpected ')'")
        return node

    def alt(self):
        branches = [self.cat()]
        while self.peek() == "|":
            self.i += 1
            branches.append(self.cat())
        return branches[0] if len(branches) == 1 else ("alt", branches)

    def cat(self):
        items = []
        while self.peek() is not None and self.peek() not in "|)":
            items.append(self.repeat())
        if not items:
            return ("empty",)
        return items[0] if len(items) == 1 else ("cat", items)

    def repeat(self):
        atom = self.atom()
        repeated = False
        while True:
            c = self.peek()
            if c == "*":
                lo, hi = 0, None
            elif c == "+":
                lo, hi = 1, None
            elif c == "?":
                lo, hi = 0, 1
            elif c == "{":
                rng = self.counted()
                if rng is None:
                    return atom
                lo, hi = rng
            else:
                return atom
            if c != "{":
                self.i += 1
            if self.peek() == "?":      # non-greedy: the same language
                self.i += 1
            if repeated:
                raise Invalid("invalid nested repetition operator")   # a** (RE2 rejects it; (a*)* is fine)
            repeated = True
            atom = ("rep", atom, lo, hi)

    def counted(self):
        """``{n}``, ``{n,}``, ``{n,m}`` at self.i, consumed; None (nothing consumed) when the brace is a literal."""
        j = self.s.find("}", self.i)
        if j < 0:
            return None
        body = self.s[self.i + 1:j]
        lo, sep, hi = body.partition(",")
        if not lo.isdigit() or (sep and hi and not hi.isdigit()) or not lo.isascii() or not hi.isascii():
            return None
        lo_n = int(lo)
        hi_n = lo_n if not sep else (None if hi == "" else int(hi))
        if lo_n > MAX_REPEAT or (hi_n is not None and (hi_n > MAX_REPEAT or hi_n < lo_n)):
            raise Invalid("bad repetition operator")
        self.i = j + 1
        return lo_n, hi_n

    def atom(self):
        c = self.s[self.i]
        self.i += 1
        if c == "(":
            if self.s.startswith("?", self.i):
                if self.s.startswith("?:", self.i):
                    self.i += 2
                elif self.s.startswith("?P<", self.i):
                    j = self.s.find(">", self.i)
                    if j < 0:
                        raise Invalid("invalid named capture")
                    self.i = j + 1
                else:
                    raise Unsupported("flags / special groups")
            node = self.alt()
            if self.peek() != ")":
                raise Invalid("missing closing )")
            self.i += 1
            return node
        if c == "[":
            return ("set", self.bracket())
        if c == ".":
            return ("set", _Set(_ASCII - {0x0A}, True))
        if c == "^":
            return ("bol",)
        if c == "$":
            return ("eol",)
        if c == "\\":
            return self.escape(in_class=False)
        if c in "*+?":
            raise Invalid("missing argument to repetition operator")
        if c == ")":
            raise Invalid("unexpected )")
        return self.literal(c)

    @staticmethod
    def literal(ch):
        if ord(ch) < 0x80:
            return ("set", _Set({ord(ch)}))
        return ("bytes", ch.encode("utf-8"))

    def escape(self, in_class):
        if self.i >= len(self.s):
            raise Invalid("trailing backslash")
        c = self.s[self.i]
        self.i += 1
        if c == "d":
            return ("set", _Set(_DIGIT))
        if c == "D":
            return ("set", _Set(_DIGIT).negate())
        if c == "w":
            return ("set", _Set(_WORD))
        if c == "W":
            return ("set", _Set(_WORD).negate())
        if c == "s":
            return ("set", _Set(_SPACE))
        if c == "S":
            return ("set", _Set(_SPACE).negate())
        if c in _ESC_LIT:
            return ("set", _Set({_ESC_LIT[c]}))
        if c == "x":
            h = self.s[self.i:self.i + 2]
            if len(h) == 2 and all(x in "0123456789abcdefABCDEF" for x in h):
                self.i += 2
                v = int(h, 16)
                if v >= 0x80:
                    raise Unsupported("non-ASCII \\x escape")
                return ("set", _Set({v}))
            raise Unsupported("\\x{...} escape")
        if not in_class:
            if c == "A":
                return ("bol",)
            if c == "z":
                return ("eol",)
            if c in "bBQEpPC":
                raise Unsupported("\\%s" % c)
        elif c in "pP":
            raise Unsupported("\\%s" % c)
        if c.isdigit():
            raise Unsupported("octal escape / backreference")
        if c.isalpha():
            raise Invalid("invalid escape sequence \\%s" % c)
        return self.literal(c)

    def bracket(self):
        neg = False
        if self.peek() == "^":
            neg = True
            self.i += 1
        out = _Set()
        first = True
        while True:
            if self.i >= len(self.s):
                raise Invalid("missing closing ]")
            c = self.s[self.i]
            if c == "]" and not first:
                self.i += 1
                break
            first = False
            if c == "[" and self.s.startswith("[:", self.i):
                raise Unsupported("POSIX class")
            lo = self.class_atom()
            if isinstance(lo, _Set):
                out = out.union(lo)
                continue
            if self.peek() == "-" and self.i + 1 < len(self.s) and self.s[self.i + 1] != "]":
                self.i += 1
                hi = self.class_atom()
                if isinstance(hi, _Set):
                    raise Invalid("bad character class range")
                if hi < lo:
                    raise Invalid("bad character class range")
                out = out.union(_Set(range(lo, hi + 1)))
            else:
                out = out.union(_Set({lo}))
        return out.negate() if neg else out

    def class_atom(self):
        """One member of a bracket class: an ASCII code point, or a _Set for ``\\d`` and friends."""
        c = self.s[self.i]
        self.i += 1
        if c == "\\":
            node = self.escape(in_class=True)
            if node[0] == "bytes":
                raise Unsupported("non-ASCII character in a class")
            st = node[1]
            if len(st.ascii) == 1 and not st.non_ascii:
                return next(iter(st.ascii))
            return st
        if ord(c) >= 0x80:
            raise Unsupported("non-ASCII character in a class")
        return ord(c)


# ---- NFA ------------------------------------------------------------------------------------------------------
class _Nfa:
    def __init__(self):
        self.eps = []      # state -> [state]
        self.bol = []      # state -> [state]   passable at offset 0 only
        self.eol = []      # state -> [state]   passable at the end only
        self.edges = []    # state -> [(frozenset of bytes, state)]

    def new(self):
        if len(self.eps) >= MAX_NFA:
            raise Unsupported("pattern too large")
        self.eps.append([]); self.bol.append([]); self.eol.append([]); self.edges.append([])
        return len(self.eps) - 1


_CONT = frozenset(range(0x80, 0xC0))


def _build(n: _Nfa, node, src):
    """Adds `node` starting at state `src`; returns its end state."""
    k = node[0]
    if k == "empty":
        return src
    if k == "bol" or k == "eol":
        dst = n.new()
        (n.bol if k == "bol" else n.eol)[src].append(dst)
        return dst
    if k == "bytes":
        cur = src
        for b in node[1]:
            nxt = n.new()
            n.edges[cur].append((frozenset({b}), nxt))
            cur = nxt
        return cur
    if k == "set":
        st, dst = node[1], n.new()
        if st.ascii:
            n.edges[src].append((st.ascii, dst))
        if st.non_ascii:   # one well-formed multi-byte sequence
            for lead, cont in ((range(0xC2, 0xE0), 1), (range(0xE0, 0xF0), 2), (range(0xF0, 0xF5), 3)):
                cur = n.new()
                n.edges[src].append((frozenset(lead), cur))
                for j in range(cont):
                    nxt = dst if j == cont - 1 else n.new()
                    n.edges[cur].append((_CONT, nxt))
                    cur = nxt
        return dst
    if k == "cat":
        cur = src
        for x in node[1]:
            cur = _build(n, x, cur)
        return cur
    if k == "alt":
        dst = n.new()
        for x in node[1]:
            s = n.new()
            n.eps[src].append(s)
            n.eps[_build(n, x, s)].append(dst)
        return dst
    if k == "rep":
        _, sub, lo, hi = node
        cur = src
        for _ in range(lo):
            cur = _build(n, sub, cur)
        if hi is None:          # sub*
            loop, dst = n.new(), n.new()
            n.eps[cur].append(loop)
            n.eps[loop].append(dst)
            n.eps[_build(n, sub, loop)].append(loop)
            return dst
        dst = n.new()
        n.eps[cur].append(dst)
        for _ in range(hi - lo):   # (sub (sub ...)?)?
            cur = _build(n, sub, cur)
            n.eps[cur].append(dst)
        return dst
    raise AssertionError(k)


class RegexDFA:
    """start state 0; ``flags[s]`` bit 0 = a match is already certain, bit 1 = a match if the text ends here."""

    def __init__(self, classmap, n_classes, trans, flags):
        self.classmap, self.n_classes, self.trans, self.flags = classmap, n_classes, trans, flags
        self.n_states = len(flags)

    def words(self):
        """u32 layout read by cbh_vm.h regex_match: [n_states, n_classes, 64 words of classmap (4 bytes each, little
        endian), flags[n_states], trans[n_states][n_classes]]."""
        out = [self.n_states, self.n_classes]
        for i in range(0, 256, 4):
            out.append(self.classmap[i] | (self.classmap[i + 1] << 8) | (self.classmap[i + 2] << 16) | (self.classmap[i + 3] << 24))
        out.extend(self.flags)
        for row in self.trans:
            out.extend(row)
        return out

    def search(self, data: bytes) -> bool:
        """The device's loop, for tests."""
        s = 0
        if self.flags[s] & 1:
            return True
        for b in data:
            s = self.trans[s][self.classmap[b]]
            if self.flags[s] & 1:
                return True
        return bool(self.flags[s] & 2)


def compile_regex(pattern: str) -> RegexDFA:
    ast = _Parser(pattern).parse()
    n = _Nfa()
    start = n.new()
    accept = _build(n, ast, start)

    def closure(states, at_start, at_end):
        seen, stack = set(states), list(states)
        while stack:
            s = stack.pop()
            for t in n.eps[s] + (n.bol[s] if at_start else []) + (n.eol[s] if at_end else []):
                if t not in seen:
                    seen.add(t)
                    stack.append(t)
        return frozenset(seen)

    # byte equivalence classes: bytes no transition set tells apart
    sigs = {}
    all_sets = {bs for edges in n.edges for bs, _ in edges}
    for b in range(256):
        sigs.setdefault(frozenset(i for i, bs in enumerate(all_sets) if b in bs), []).append(b)
    classmap, reps = [0] * 256, []
    for cls, (sig, bs) in enumerate(sorted(sigs.items(), key=lambda kv: kv[1][0])):
        reps.append(bs[0])
        for b in bs:
            classmap[b] = cls

    restart = closure({start}, False, False)   # a match may begin at any later offset (no ^ there)
    d0 = closure({start}, True, False)
    # state 0 is "nothing consumed yet" and only that: a later position with the same NFA set gets its own state,
    # because at offset 0 an empty text is at its start AND its end (`^$`, `$^`)
    index, order, trans, flags = {}, [d0], [], []
    k = 0
    while k < len(order):
        cur = order[k]
        k += 1
        certain = accept in cur
        at_end = accept in closure(cur, k == 1, True)
        flags.append((1 if certain else 0) | (2 if (certain or at_end) else 0))
        row = []
        for rb in reps:
            if certain:          # absorbing: the answer is known
                row.append(k - 1)
                continue
            moved = {t for s in cur for bs, t in n.edges[s] if rb in bs}
            nxt = closure(moved, False, False) | restart
            if nxt not in index:
                if len(order) >= MAX_STATES:
                    raise Unsupported("automaton beyond %d states" % MAX_STATES)
                index[nxt] = len(order)
                order.append(nxt)
            row.append(index[nxt])
        trans.append(row)
    return RegexDFA(classmap, len(reps), trans, flags)
