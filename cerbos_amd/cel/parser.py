"""CEL (Common Expression Language) text -> AST.

The reference compiles condition text with github.com/google/cel-go v0.30.0
(``internal/conditions/cel.go:170-176``; not vendored under /root/reference).
This is a from-scratch recursive-descent parser for the published CEL grammar
(cel-spec ``doc/langdef.md``), with the macros the reference enables
(``cel.go:65-88``: standard macros, ``ext.Bindings``, ``ext.TwoVarComprehensions``).

AST nodes are plain tuples (hashable, trivially serialisable):

  ('lit', kind, value)      kind in null|bool|int|uint|double|string|bytes
  ('ident', name)
  ('select', operand, field)
  ('has', operand, field)           has(operand.field)
  ('index', operand, index)
  ('call', name, target|None, args) function / method call
  ('list', elems) ; ('map', ((k, v), ...))
  ('not', x) ; ('neg', x)
  ('bin', op, a, b)         op in == != < <= > >= in + - * / %
  ('and', a, b) ; ('or', a, b) ; ('tern', c, a, b)
  ('comp', kind, target, vars, args)   comprehension macros
  ('bind', var, init, body)            cel.bind(var, init, body)
"""
from __future__ import annotations


class CELSyntaxError(ValueError):
    pass


_PUNCT3 = ()
_PUNCT2 = ("&&", "||", "==", "!=", "<=", ">=", ".?", "[?")
_PUNCT1 = "()[]{}.,?:+-*/%!<>="

_COMP_MACROS = {
    # name: allowed arg counts
    "all": (2, 3),
    "exists": (2, 3),
    "exists_one": (2, 3),
    "existsOne": (2, 3),
    "map": (2, 3),
    "filter": (2,),
    "transformList": (3, 4),
    "transformMap": (3, 4),
    "transformMapEntry": (3, 4),
    "sortBy": (2,),
}

_ESC = {"a": "\a", "b": "\b", "f": "\f", "n": "\n", "r": "\r", "t": "\t", "v": "\v",
        "\\": "\\", "'": "'", '"': '"', "`": "`", "?": "?"}


def _unescape(body: str, is_bytes: bool):
    out = bytearray() if is_bytes else []
    i, n = 0, len(body)

    def emit_cp(cp):
        if is_bytes:
            out.extend(chr(cp).encode("utf-8"))
        else:
            out.append(chr(cp))

    while i < n:
        c = body[i]
        if c != "\\":
            emit_cp(ord(c))
            i += 1
            continue
        i += 1
        if i >= n:
            raise CELSyntaxError("dangling escape")
        e = body[i]
        if e in _ESC:
            emit_cp(ord(_ESC[e]))
            i += 1
        elif e in "xX":
            v = int(body[i + 1:i + 3], 16)
            if is_bytes:
                out.append(v)
            else:
                out.append(chr(v))
            i += 3
        elif e == "u":
            emit_cp(int(body[i + 1:i + 5], 16))
            i += 5
        elif e == "U":
            emit_cp(int(body[i + 1:i + 9], 16))
            i += 9
        elif e in "0123":
            v = int(body[i:i + 3], 8)
            if is_bytes:
                out.append(v)
            else:
                out.append(chr(v))
            i += 3
        else:
            raise CELSyntaxError(f"bad escape \\{e}")
    return bytes(out) if is_bytes else "".join(out)


def tokenize(src: str):
    toks = []
    i, n = 0, len(src)
    while i < n:
        c = src[i]
        if c in " \t\r\n\f":
            i += 1
            continue
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j + 1
            continue
        # string / bytes literals (with optional r / b prefixes)
        j = i
        raw = is_bytes = False
        while j < n and src[j] in "rRbB" and j - i < 2:
            j += 1
        if j < n and src[j] in "\"'" and all(ch in "rRbB" for ch in src[i:j]):
            prefix = src[i:j].lower()
            if len(set(prefix)) == len(prefix):
                raw = "r" in prefix
                is_bytes = "b" in prefix
                q = src[j]
                if src.startswith(q * 3, j):
                    end = src.find(q * 3, j + 3)
                    if end < 0:
                        raise CELSyntaxError("unterminated string")
                    body = src[j + 3:end]
                    nxt = end + 3
                else:
                    k = j + 1
                    while k < n and src[k] != q:
                        if src[k] == "\\" and not raw:
                            k += 1
                        if k < n and src[k] == "\n":
                            raise CELSyntaxError("newline in string")
                        k += 1
                    if k >= n:
                        raise CELSyntaxError("unterminated string")
                    body = src[j + 1:k]
                    nxt = k + 1
                if raw:
                    val = body.encode("utf-8") if is_bytes else body
                else:
                    val = _unescape(body, is_bytes)
                toks.append(("bytes" if is_bytes else "string", val))
                i = nxt
                continue
        if c.isdigit() or (c == "." and i + 1 < n and src[i + 1].isdigit()):
            j = i
            if src.startswith(("0x", "0X"), i):
                j = i + 2
                while j < n and src[j] in "0123456789abcdefABCDEF":
                    j += 1
                text = src[i:j]
                if j < n and src[j] in "uU":
                    toks.append(("uint", int(text, 16)))
                    j += 1
                else:
                    toks.append(("int", int(text, 16)))
                i = j
                continue
            while j < n and src[j].isdigit():
                j += 1
            is_float = False
            if j < n and src[j] == "." and j + 1 < n and src[j + 1].isdigit():
                is_float = True
                j += 1
                while j < n and src[j].isdigit():
                    j += 1
            if j < n and src[j] in "eE":
                k = j + 1
                if k < n and src[k] in "+-":
                    k += 1
                if k < n and src[k].isdigit():
                    is_float = True
                    j = k
                    while j < n and src[j].isdigit():
                        j += 1
            text = src[i:j]
            if is_float:
                toks.append(("double", float(text)))
            elif j < n and src[j] in "uU":
                toks.append(("uint", int(text)))
                j += 1
            else:
                toks.append(("int", int(text)))
            i = j
            continue
        if c.isalpha() or c == "_":
            j = i + 1
            while j < n and (src[j].isalnum() or src[j] == "_"):
                j += 1
            toks.append(("ident", src[i:j]))
            i = j
            continue
        two = src[i:i + 2]
        if two in _PUNCT2:
            toks.append(("p", two))
            i += 2
            continue
        if c in _PUNCT1:
            toks.append(("p", c))
            i += 1
            continue
        raise CELSyntaxError(f"unexpected character {c!r} at {i}")
    toks.append(("eof", None))
    return toks


class _Parser:
    def __init__(self, src: str):
        self.toks = tokenize(src)
        self.i = 0

    def peek(self):
        return self.toks[self.i]

    def at(self, p):
        t = self.toks[self.i]
        return t[0] == "p" and t[1] == p

    def accept(self, p):
        if self.at(p):
            self.i += 1
            return True
        return False

    def expect(self, p):
        if not self.accept(p):
            raise CELSyntaxError(f"expected {p!r}, got {self.peek()!r}")

    def parse(self):
        e = self.expr()
        if self.peek()[0] != "eof":
            raise CELSyntaxError(f"unexpected token {self.peek()!r}")
        return e

    def expr(self):
        c = self.cond_or()
        if self.accept("?"):
            a = self.cond_or()
            self.expect(":")
            b = self.expr()
            return ("tern", c, a, b)
        return c

    def cond_or(self):
        e = self.cond_and()
        while self.accept("||"):
            e = ("or", e, self.cond_and())
        return e

    def cond_and(self):
        e = self.relation()
        while self.accept("&&"):
            e = ("and", e, self.relation())
        return e

    def relation(self):
        e = self.addition()
        while True:
            t = self.peek()
            if t[0] == "p" and t[1] in ("<", "<=", ">", ">=", "==", "!="):
                self.i += 1
                e = ("bin", t[1], e, self.addition())
            elif t[0] == "ident" and t[1] == "in":
                self.i += 1
                e = ("bin", "in", e, self.addition())
            else:
                return e

    def addition(self):
        e = self.multiplication()
        while True:
            if self.accept("+"):
                e = ("bin", "+", e, self.multiplication())
            elif self.accept("-"):
                e = ("bin", "-", e, self.multiplication())
            else:
                return e

    def multiplication(self):
        e = self.unary()
        while True:
            if self.accept("*"):
                e = ("bin", "*", e, self.unary())
            elif self.accept("/"):
                e = ("bin", "/", e, self.unary())
            elif self.accept("%"):
                e = ("bin", "%", e, self.unary())
            else:
                return e

    def unary(self):
        if self.at("!"):
            k = 0
            while self.accept("!"):
                k += 1
            e = self.member()
            for _ in range(k):
                e = ("not", e)
            return e
        if self.at("-"):
            k = 0
            while self.accept("-"):
                k += 1
            e = self.member()
            # cel-go folds a minus sign straight into numeric literals
            if e[0] == "lit" and e[1] in ("int", "double") and k % 2 == 1:
                e = ("lit", e[1], -e[2])
                k -= 1
            for _ in range(k):
                e = ("neg", e)
            return e
        return self.member()

    def member(self):
        e = self.primary()
        while True:
            if self.accept("."):
                t = self.peek()
                if t[0] != "ident":
                    raise CELSyntaxError(f"expected field name, got {t!r}")
                self.i += 1
                name = t[1]
                if self.accept("("):
                    args = self.args(")")
                    e = self.method(e, name, args)
                else:
                    e = ("select", e, name)
            elif self.accept("["):
                idx = self.expr()
                self.expect("]")
                e = ("index", e, idx)
            elif self.at(".?") or self.at("[?"):
                raise CELSyntaxError("optional field selection is not supported")
            else:
                return e

    def args(self, close):
        out = []
        if self.accept(close):
            return out
        while True:
            out.append(self.expr())
            if self.accept(","):
                if self.at(close):  # trailing comma
                    self.i += 1
                    return out
                continue
            self.expect(close)
            return out

    def method(self, target, name, args):
        if name in _COMP_MACROS and len(args) in _COMP_MACROS[name]:
            nvars = 1
            if name in ("all", "exists", "exists_one", "existsOne") and len(args) == 3:
                nvars = 2
            elif name in ("transformList", "transformMap", "transformMapEntry"):
                nvars = 2
            ok = all(a[0] == "ident" for a in args[:nvars])
            if ok:
                vars_ = tuple(a[1] for a in args[:nvars])
                return ("comp", name, target, vars_, tuple(args[nvars:]))
        # namespaced functions parse as method calls on an identifier: keep them as such;
        # the evaluator resolves e.g. ('call','contains',('ident','sets'),...) .
        if name == "bind" and target == ("ident", "cel") and len(args) == 3 and args[0][0] == "ident":
            return ("bind", args[0][1], args[1], args[2])
        return ("call", name, target, tuple(args))

    def primary(self):
        t = self.peek()
        k = t[0]
        if k in ("int", "uint", "double", "string", "bytes"):
            self.i += 1
            return ("lit", k, t[1])
        if k == "ident":
            self.i += 1
            name = t[1]
            if name == "true":
                return ("lit", "bool", True)
            if name == "false":
                return ("lit", "bool", False)
            if name == "null":
                return ("lit", "null", None)
            if self.accept("("):
                args = self.args(")")
                if name == "has":
                    if len(args) != 1 or args[0][0] != "select":
                        raise CELSyntaxError("invalid argument to has() macro")
                    return ("has", args[0][1], args[0][2])
                return ("call", name, None, tuple(args))
            return ("ident", name)
        if k == "p":
            if t[1] == "(":
                self.i += 1
                e = self.expr()
                self.expect(")")
                return e
            if t[1] == "[":
                self.i += 1
                return ("list", tuple(self.args("]")))
            if t[1] == "{":
                self.i += 1
                entries = []
                if not self.accept("}"):
                    while True:
                        kx = self.expr()
                        self.expect(":")
                        vx = self.expr()
                        entries.append((kx, vx))
                        if self.accept(","):
                            if self.accept("}"):
                                break
                            continue
                        self.expect("}")
                        break
                return ("map", tuple(entries))
            if t[1] == ".":
                # leading dot = root-namespace identifier; same thing for our purposes
                self.i += 1
                return self.primary()
        raise CELSyntaxError(f"unexpected token {t!r}")


_CACHE: dict = {}


def parse(src: str):
    """Parse CEL source text to an AST (cached by text, like the reference's
    ``ProgramCache`` keyed by ``Expr.Original``, ruletable.go:518-559)."""
    ast = _CACHE.get(src)
    if ast is None:
        ast = _Parser(src).parse()
        _CACHE[src] = ast
    return ast


def walk(ast):
    """Pre-order iteration over all nodes."""
    stack = [ast]
    while stack:
        n = stack.pop()
        if not isinstance(n, tuple) or not n or not isinstance(n[0], str):
            continue
        yield n
        k = n[0]
        if k in ("lit", "ident"):
            continue
        if k in ("select", "has"):
            stack.append(n[1])
        elif k == "index":
            stack.extend((n[2], n[1]))
        elif k == "call":
            stack.extend(reversed(n[3]))
            if n[2] is not None:
                stack.append(n[2])
        elif k == "list":
            stack.extend(reversed(n[1]))
        elif k == "map":
            for kk, vv in reversed(n[1]):
                stack.extend((vv, kk))
        elif k in ("not", "neg"):
            stack.append(n[1])
        elif k == "bin":
            stack.extend((n[3], n[2]))
        elif k in ("and", "or"):
            stack.extend((n[2], n[1]))
        elif k == "tern":
            stack.extend((n[3], n[2], n[1]))
        elif k == "comp":
            stack.extend(reversed(n[4]))
            stack.append(n[2])
        elif k == "bind":
            stack.extend((n[3], n[2]))
