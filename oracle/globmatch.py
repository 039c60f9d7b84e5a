"""ORACLE (test infrastructure only - never imported by the product path).

Glob matching as used by the reference's index dimensions and role-policy allow-lists.
The algorithm lives in github.com/gobwas/glob v0.2.3 (go.mod:41; NOT under
/root/reference). This restates its documented pattern language with separator ':'
(``internal/util/globs_common.go:31``: ``glob.Compile(expr, ':')``):

  *        any run of non-separator characters (possibly empty)
  **       any run of characters including separators
  ?        exactly one non-separator character
  [abc] [a-z] [!abc] [!a-z]   one character in / not in the class
  {a,b}    alternatives (may nest)
  \\x       literal x

A bare ``*`` is rewritten to ``**`` first (``globs_common.go:74-81``).
Parity pinning: the reference has no direct glob unit test; ``*``, ``**`` and ``prefix:*``
are pinned indirectly by engine golden cases 00/04/07/14-16 and
``index/index_test.go:1098-1200``; ``? [] {}`` are parity-unpinned.
"""
from __future__ import annotations

import functools
import re

SEP = ":"


def fix_glob(g: str) -> str:
    return "**" if g == "*" else g


def _translate(pat: str, i: int, closers: str):
    """Translate pat[i:] up to (not including) an unescaped char in ``closers`` at depth 0."""
    out = []
    n = len(pat)
    while i < n:
        c = pat[i]
        if c in closers:
            break
        if c == "\\":
            if i + 1 >= n:
                raise ValueError("dangling escape in glob")
            out.append(re.escape(pat[i + 1]))
            i += 2
        elif c == "*":
            if i + 1 < n and pat[i + 1] == "*":
                out.append(r"[\s\S]*")
                i += 2
            else:
                out.append("[^%s]*" % re.escape(SEP))
                i += 1
        elif c == "?":
            out.append("[^%s]" % re.escape(SEP))
            i += 1
        elif c == "[":
            j = i + 1
            neg = j < n and pat[j] == "!"
            if neg:
                j += 1
            k = pat.find("]", j)
            if k < 0:
                raise ValueError("unterminated class in glob")
            body = pat[j:k]
            cls = []
            t = 0
            while t < len(body):
                if t + 2 < len(body) and body[t + 1] == "-":
                    cls.append("%s-%s" % (re.escape(body[t]), re.escape(body[t + 2])))
                    t += 3
                else:
                    cls.append(re.escape(body[t]))
                    t += 1
            out.append("[%s%s]" % ("^" if neg else "", "".join(cls)))
            i = k + 1
        elif c == "{":
            alts = []
            i += 1
            while True:
                sub, i = _translate(pat, i, ",}")
                alts.append(sub)
                if i >= n:
                    raise ValueError("unterminated alternatives in glob")
                if pat[i] == "}":
                    i += 1
                    break
                i += 1  # ','
            out.append("(?:%s)" % "|".join(alts))
        else:
            out.append(re.escape(c))
            i += 1
    return "".join(out), i


@functools.lru_cache(maxsize=4096)
def compile_glob(pattern: str):
    try:
        rx, i = _translate(pattern, 0, "")
        return re.compile(rx + r"\Z")
    except (ValueError, re.error):
        return None  # invalid glob never matches (globs_common.go:33-36)


def glob_match(pattern: str, value: str) -> bool:
    """``g.Match(value)`` for a pattern already passed through fix_glob."""
    rx = compile_glob(pattern)
    return bool(rx and rx.match(value))


def matches_glob(g: str, value: str) -> bool:
    """``util.MatchesGlob`` (globs_common.go:42-44)."""
    return glob_match(fix_glob(g), value)
