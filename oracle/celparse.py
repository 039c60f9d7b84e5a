"""TEST INFRASTRUCTURE ONLY (the oracle's own CEL reader; nothing under cerbos_amd/ imports it).

CEL text -> AST for oracle/celeval.py, written apart from the product's parser (cerbos_amd/cel/parser.py: hand-rolled
scanner + one recursive function per grammar level) so that a parser bug is not common mode between the product and its
checker: a regular-expression scanner and a precedence-climbing (Pratt) expression parser over the published CEL grammar
(cel-spec doc/langdef.md; what cel-go v0.30.0 - go.mod:44, not vendored - parses for internal/conditions/cel.go:170-176),
with the macros the reference enables (cel.go:65-88: the standard macros, ext.Bindings, ext.TwoVarComprehensions).
tests/test_oracle_parser.py diffs the two parsers' trees over every expression the goldens and the generators hold.

The tree is the same plain-tuple form both evaluators read:
  ('lit', kind, value) ('ident', name) ('select', x, field) ('has', x, field) ('index', x, i) ('call', name, target|None, args)
  ('list', elems) ('map', ((k, v), ...)) ('not', x) ('neg', x) ('bin', op, a, b) ('and', a, b) ('or', a, b) ('tern', c, a, b)
  ('comp', macro, target, vars, args) ('bind', var, init, body)
"""
from __future__ import annotations

import re


class CelParseError(ValueError):
    pass


_TOKEN = re.compile(r"""
    (?P<ws>[ \t\r\n\f]+|//[^\n]*)
  | (?P<str>(?i:rb|br|r|b)?(?:\"\"\"(?:.|\n)*?\"\"\"|'''(?:.|\n)*?'''|"(?:\\.|[^"\\\n])*"|'(?:\\.|[^'\\\n])*'))
  | (?P<hex>0[xX][0-9a-fA-F]+[uU]?)
  | (?P<flt>(?:\d+\.\d+(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+|\.\d+(?:[eE][+-]?\d+)?))
  | (?P<int>\d+[uU]?)
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<op>&&|\|\||==|!=|<=|>=|\.\?|\[\?|[()\[\]{}.,?:+\-*/%!<>=])
""", re.X)
# a raw string's prefix decides how its backslashes read, so raw strings are scanned by a pattern of their own
_RAW = re.compile(r"""(?i:rb|br|r)(?:\"\"\"(?:.|\n)*?\"\"\"|'''(?:.|\n)*?'''|"[^"\n]*"|'[^'\n]*')""")

_SIMPLE_ESCAPES = {"a": 7, "b": 8, "f": 12, "n": 10, "r": 13, "t": 9, "v": 11, "\\": 92, "'": 39, '"': 34, "`": 96, "?": 63}


def _decode(body, as_bytes):
    """Escape sequences of a non-raw literal: in a string \\x.. \\ooo are code points, in bytes they are octets."""
    out = bytearray() if as_bytes else []

    def put_cp(cp):
        if as_bytes:
            out.extend(chr(cp).encode("utf-8"))
        else:
            out.append(chr(cp))

    def put_octet(v):
        if as_bytes:
            out.append(v)
        else:
            out.append(chr(v))
    pos = 0
    for m in re.finditer(r"\\(?:([abfnrtv\\'\"`?])|[xX]([0-9a-fA-F]{2})|u([0-9a-fA-F]{4})|U([0-9a-fA-F]{8})|([0-3][0-7]{2})|(.|$))", body, re.S):
        for ch in body[pos:m.start()]:
            put_cp(ord(ch))
        pos = m.end()
        simple, hx, u4, u8, octal, bad = m.groups()
        if simple is not None:
            put_cp(_SIMPLE_ESCAPES[simple])
        elif hx is not None:
            put_octet(int(hx, 16))
        elif u4 is not None:
            put_cp(int(u4, 16))
        elif u8 is not None:
            put_cp(int(u8, 16))
        elif octal is not None:
            put_octet(int(octal, 8))
        else:
            raise CelParseError("bad escape \\%s" % (bad or ""))
    for ch in body[pos:]:
        put_cp(ord(ch))
    return bytes(out) if as_bytes else "".join(out)


def _string_token(text):
    i = 0
    while text[i] not in "\"'":
        i += 1
    prefix = text[:i].lower()
    raw, as_bytes = "r" in prefix, "b" in prefix
    q = text[i]
    body = text[i + 3:-3] if text.startswith(q * 3, i) and len(text) - i >= 6 else text[i + 1:-1]
    if raw:
        return ("bytes", body.encode("utf-8")) if as_bytes else ("string", body)
    return ("bytes" if as_bytes else "string", _decode(body, as_bytes))


def scan(src):
    toks, pos, n = [], 0, len(src)
    while pos < n:
        m = _RAW.match(src, pos)
        if m:
            toks.append(_string_token(m.group(0)))
            pos = m.end()
            continue
        m = _TOKEN.match(src, pos)
        if not m:
            if src[pos] in "\"'":
                raise CelParseError("unterminated string")
            raise CelParseError("unexpected character %r at %d" % (src[pos], pos))
        pos = m.end()
        kind = m.lastgroup
        text = m.group(kind)
        if kind == "ws":
            continue
        if kind == "str":
            toks.append(_string_token(text))
        elif kind == "hex":
            toks.append(("uint", int(text[:-1], 16)) if text[-1] in "uU" else ("int", int(text, 16)))
        elif kind == "flt":
            toks.append(("double", float(text)))
        elif kind == "int":
            toks.append(("uint", int(text[:-1])) if text[-1] in "uU" else ("int", int(text)))
        elif kind == "id":
            toks.append(("id", text))
        else:
            toks.append(("op", text))
    toks.append(("end", None))
    return toks


# binary operators: binding power (higher binds tighter); all left-associative (langdef.md: ||, &&, relations, + -, * / %)
_BINARY = {"||": 1, "&&": 2, "<": 3, "<=": 3, ">": 3, ">=": 3, "==": 3, "!=": 3, "in": 3, "+": 4, "-": 4, "*": 5, "/": 5, "%": 5}
_MACRO_ARGS = {"all": (2, 3), "exists": (2, 3), "exists_one": (2, 3), "existsOne": (2, 3), "map": (2, 3), "filter": (2,),
               "transformList": (3, 4), "transformMap": (3, 4), "transformMapEntry": (3, 4), "sortBy": (2,)}
_TWO_VAR_WHEN_THREE = ("all", "exists", "exists_one", "existsOne")
_ALWAYS_TWO_VARS = ("transformList", "transformMap", "transformMapEntry")


class _Pratt:
    def __init__(self, src):
        self.t = scan(src)
        self.p = 0

    def cur(self):
        return self.t[self.p]

    def is_op(self, text):
        return self.t[self.p] == ("op", text)

    def take(self, text):
        if self.is_op(text):
            self.p += 1
            return True
        return False

    def need(self, text):
        if not self.take(text):
            raise CelParseError("expected %r, found %r" % (text, self.cur()))

    # expr : conditionalOr ('?' conditionalOr ':' expr)?
    def expression(self):
        cond = self.binary(1)
        if self.take("?"):
            then = self.binary(1)
            self.need(":")
            return ("tern", cond, then, self.expression())
        return cond

    def _peek_binary(self):
        k, v = self.cur()
        if k == "op" and v in _BINARY:
            return v
        if k == "id" and v == "in":
            return "in"
        return None

    def binary(self, min_power):
        left = self.unary()
        while True:
            op = self._peek_binary()
            if op is None or _BINARY[op] < min_power:
                return left
            self.p += 1
            right = self.binary(_BINARY[op] + 1)
            left = ("or", left, right) if op == "||" else ("and", left, right) if op == "&&" else ("bin", op, left, right)

    # unary : member | '!'+ member | '-'+ member
    def unary(self):
        for sign, node in (("!", "not"), ("-", "neg")):
            if self.is_op(sign):
                count = 0
                while self.take(sign):
                    count += 1
                operand = self.member()
                if sign == "-" and operand[0] == "lit" and operand[1] in ("int", "double") and count & 1:
                    operand = ("lit", operand[1], -operand[2])     # the sign belongs to a numeric literal (cel-go folds it)
                    count -= 1
                for _ in range(count):
                    operand = (node, operand)
                return operand
        return self.member()

    def member(self):
        node = self.primary()
        while True:
            if self.take("."):
                k, name = self.cur()
                if k != "id":
                    raise CelParseError("expected a field name, found %r" % (self.cur(),))
                self.p += 1
                node = self.method(node, name, self.arguments(")")) if self.take("(") else ("select", node, name)
            elif self.take("["):
                index = self.expression()
                self.need("]")
                node = ("index", node, index)
            elif self.is_op(".?") or self.is_op("[?"):
                raise CelParseError("optional field selection is not supported")
            else:
                return node

    def arguments(self, closer):
        items = []
        while not self.take(closer):
            items.append(self.expression())
            if not self.take(","):
                self.need(closer)
                break
        return items

    def method(self, target, name, args):
        if name in _MACRO_ARGS and len(args) in _MACRO_ARGS[name]:
            n_vars = 2 if (name in _ALWAYS_TWO_VARS or (name in _TWO_VAR_WHEN_THREE and len(args) == 3)) else 1
            if all(a[0] == "ident" for a in args[:n_vars]):
                return ("comp", name, target, tuple(a[1] for a in args[:n_vars]), tuple(args[n_vars:]))
        if name == "bind" and target == ("ident", "cel") and len(args) == 3 and args[0][0] == "ident":
            return ("bind", args[0][1], args[1], args[2])
        return ("call", name, target, tuple(args))

    def primary(self):
        k, v = self.cur()
        if k in ("int", "uint", "double", "string", "bytes"):
            self.p += 1
            return ("lit", k, v)
        if k == "id":
            self.p += 1
            if v in ("true", "false"):
                return ("lit", "bool", v == "true")
            if v == "null":
                return ("lit", "null", None)
            if self.take("("):
                args = self.arguments(")")
                if v == "has":
                    if len(args) != 1 or args[0][0] != "select":
                        raise CelParseError("invalid argument to has() macro")
                    return ("has", args[0][1], args[0][2])
                return ("call", v, None, tuple(args))
            return ("ident", v)
        if k == "op":
            if v == "(":
                self.p += 1
                inner = self.expression()
                self.need(")")
                return inner
            if v == "[":
                self.p += 1
                return ("list", tuple(self.arguments("]")))
            if v == "{":
                self.p += 1
                entries = []
                while not self.take("}"):
                    key = self.expression()
                    self.need(":")
                    entries.append((key, self.expression()))
                    if not self.take(","):
                        self.need("}")
                        break
                return ("map", tuple(entries))
            if v == ".":       # a leading dot names the root namespace
                self.p += 1
                return self.primary()
        raise CelParseError("unexpected token %r" % ((k, v),))


_MEMO: dict = {}


def parse(src: str):
    tree = _MEMO.get(src)
    if tree is None:
        p = _Pratt(src)
        tree = p.expression()
        if p.cur()[0] != "end":
            raise CelParseError("unexpected token %r" % (p.cur(),))
        _MEMO[src] = tree
    return tree
