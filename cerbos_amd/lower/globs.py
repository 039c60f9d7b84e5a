"""Glob patterns -> bit-parallel NFA tables for the device (and a host simulation of the
same automaton used at lowering time for the table's own strings).

Pattern language: github.com/gobwas/glob v0.2.3 with separator ':' as the reference
compiles it (``internal/util/globs_common.go:31``); a bare ``*`` means ``**``
(``globs_common.go:74-81``).  See SURVEY.md Appendix C.

Each pattern is expanded (``{a,b}`` alternatives) into linear element sequences.  A linear
pattern with n elements owns n+1 consecutive state bits ("position before element i");
element kinds:

  BYTES(set)          consume one byte in ``set`` and advance
  STAR(set)           self-loop on bytes in ``set``; may also match empty (epsilon to i+1)

``?`` and negated classes consume one code point: the lead byte advances, UTF-8 continuation
bytes (0x80-0xBF) then self-loop on the following position.

Device step for input byte c (cbh_engine.hip, cbh_resolve_globs_kernel):
    A = ((A & cls[c]) << 1) | (A & self[c]);  A |= closure over star positions.
"""
from __future__ import annotations

SEP = ord(":")
_CONT = frozenset(range(0x80, 0xC0))
_ALL = frozenset(range(256))
_NONSEP = _ALL - {SEP}
_LEAD_NONSEP = _NONSEP - _CONT

META_CHARS = set("*?[{\\")


class GlobError(ValueError):
    pass


def fix_glob(g: str) -> str:
    return "**" if g == "*" else g


def has_meta(s: str) -> bool:
    return any(ch in META_CHARS for ch in s)


# element = (kind, byteset, cont_after)   kind: 'B' | 'S'
def _parse_seq(pat: str, i: int, closers: str):
    """Returns (list of alternatives-expanded sequences, next index)."""
    seqs = [[]]
    n = len(pat)
    while i < n:
        c = pat[i]
        if c in closers:
            break
        if c == "\\":
            if i + 1 >= n:
                raise GlobError("dangling escape")
            for b in pat[i + 1].encode("utf-8"):
                for s in seqs:
                    s.append(("B", frozenset([b]), False))
            i += 2
        elif c == "*":
            if i + 1 < n and pat[i + 1] == "*":
                el = ("S", _ALL, False)
                i += 2
            else:
                el = ("S", _NONSEP, False)
                i += 1
            for s in seqs:
                s.append(el)
        elif c == "?":
            for s in seqs:
                s.append(("B", _LEAD_NONSEP, True))
            i += 1
        elif c == "[":
            j = i + 1
            neg = j < n and pat[j] == "!"
            if neg:
                j += 1
            k = pat.find("]", j)
            if k < 0:
                raise GlobError("unterminated character class")
            body = pat[j:k]
            members = set()
            t = 0
            while t < len(body):
                if t + 2 < len(body) and body[t + 1] == "-":
                    lo, hi = ord(body[t]), ord(body[t + 2])
                    if hi > 0x7F or lo > 0x7F:
                        raise GlobError("non-ASCII character class is not supported")
                    members.update(range(lo, hi + 1))
                    t += 3
                else:
                    if ord(body[t]) > 0x7F:
                        raise GlobError("non-ASCII character class is not supported")
                    members.add(ord(body[t]))
                    t += 1
            if neg:
                el = ("B", frozenset((_ALL - _CONT) - members), True)
            else:
                el = ("B", frozenset(members), False)
            for s in seqs:
                s.append(el)
            i = k + 1
        elif c == "{":
            i += 1
            alts = []
            while True:
                sub, i = _parse_seq(pat, i, ",}")
                alts.extend(sub)
                if i >= n:
                    raise GlobError("unterminated alternatives")
                if pat[i] == "}":
                    i += 1
                    break
                i += 1
            seqs = [s + a for s in seqs for a in alts]
        else:
            for b in c.encode("utf-8"):
                for s in seqs:
                    s.append(("B", frozenset([b]), False))
            i += 1
    return seqs, i


def parse_glob(pattern: str):
    """Pattern text (already through fix_glob) -> list of linear element sequences."""
    seqs, i = _parse_seq(pattern, 0, "")
    if i != len(pattern):
        raise GlobError("unexpected %r" % pattern[i])
    return seqs


class GlobNFA:
    """Union automaton of the glob patterns of one index dimension."""

    MAX_GLOBS = 64
    MAX_WORDS = 8

    def __init__(self, patterns):
        self.patterns = list(patterns)  # glob index = position in this list
        if len(self.patterns) > self.MAX_GLOBS:
            raise GlobError("more than %d glob patterns in one dimension" % self.MAX_GLOBS)
        self.nbits = 0
        self.init = 0
        self.star = 0
        self.cls = [0] * 256
        self.self_ = [0] * 256
        self.accept = []  # (bit, glob index)
        self.valid = []
        for gi, pat in enumerate(self.patterns):
            try:
                seqs = parse_glob(fix_glob(pat))
            except GlobError:
                self.valid.append(False)  # an invalid glob never matches (globs_common.go:33-36)
                continue
            self.valid.append(True)
            for seq in seqs:
                base = self.nbits
                self.init |= 1 << base
                for k, (kind, bset, cont_after) in enumerate(seq):
                    bit = 1 << (base + k)
                    if kind == "B":
                        for b in bset:
                            self.cls[b] |= bit
                        if cont_after:
                            for b in _CONT:
                                self.self_[b] |= bit << 1
                    else:
                        self.star |= bit
                        for b in bset:
                            self.self_[b] |= bit
                self.accept.append((base + len(seq), gi))
                self.nbits += len(seq) + 1
        self.words = (self.nbits + 63) // 64
        if self.words > self.MAX_WORDS:
            raise GlobError("glob automaton needs %d state words (max %d)" % (self.words, self.MAX_WORDS))
        self.init = self._closure(self.init)

    def _closure(self, a: int) -> int:
        while True:
            nx = a | ((a & self.star) << 1)
            if nx == a:
                return a
            a = nx

    def match_bits(self, data: bytes) -> int:
        """Host simulation of exactly the device automaton: bitmask over glob indices."""
        if not self.patterns:
            return 0
        a = self.init
        for c in data:
            a = ((a & self.cls[c]) << 1) | (a & self.self_[c])
            a = self._closure(a)
        out = 0
        for bit, gi in self.accept:
            if (a >> bit) & 1:
                out |= 1 << gi
        return out

    def tables(self):
        """Serialised section: init[NW] star[NW] cls[256][NW] self[256][NW] (u64),
        then u32 n_accept, u32 pad, (bit, glob index) pairs."""
        import numpy as np
        nw = self.words
        if nw == 0:
            return b""
        mask = (1 << 64) - 1

        def words(v):
            return [(v >> (64 * w)) & mask for w in range(nw)]

        arr = []
        arr.extend(words(self.init))
        arr.extend(words(self.star))
        for c in range(256):
            arr.extend(words(self.cls[c]))
        for c in range(256):
            arr.extend(words(self.self_[c]))
        u64 = np.array(arr, dtype=np.uint64).tobytes()
        acc = [len(self.accept), 0]
        for bit, gi in self.accept:
            acc.extend((bit, gi))
        return u64 + np.array(acc, dtype=np.uint32).tobytes()
