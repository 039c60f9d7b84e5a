"""cerbos_amd/cel/fold.py (the lowering's constant folder) held against oracle/celeval.py - the CEL restatement pinned on the
reference's known-answer tests - and against those tests directly:

* every closed expression of TestCerbosLib (cerbos_lib_test.go) and every closed leaf of cel_eval/*.yaml that the folder
  turns into a literal must fold to what the reference asserts (true), and an expression the reference expects to FAIL
  must be left unfolded (the device raises the error);
* generated constant expressions over the folded function families: whenever the folder produces a literal the oracle
  must compute the same value, and what the oracle rejects with a CEL error the folder must leave alone."""
import math
import random

import pytest

from cerbos_amd.cel import fold as F
from cerbos_amd.cel import parser
from helpers import load_json
from oracle import celeval

NOW = 1_700_000_000_000_000_000
LIB = load_json("cerbos_lib_kats.json")["cases"]
KATS = load_json("cel_eval_cases.json")


def _folded(expr):
    out = F.fold(parser.parse(expr))
    return out if out[0] in ("lit", "list", "map") and F._const_node(out) else None


def _lit_value(n):
    if n[0] == "lit":
        return celeval.UInt(n[2]) if n[1] == "uint" else n[2]
    if n[0] == "list":
        return [_lit_value(e) for e in n[1]]
    return {_lit_value(k): _lit_value(v) for k, v in n[1]}


def test_library_kats_fold_to_what_the_reference_asserts():
    done = 0
    for c in LIB:
        lit = _folded(c["expr"])
        if c["wantErr"]:
            assert lit is None, c["expr"]
            continue
        if lit is not None:
            assert lit == ("lit", "bool", True), c["expr"]
            done += 1
    assert done >= 135, done   # (what reads the clock - now(), timeSince - is not folded)


def test_closed_kat_leaves_fold_to_true():
    done = 0
    for case in KATS:
        if case["want"] is not True or "all" not in case["condition"]:
            continue
        for m in case["condition"]["all"]["of"]:
            if "expr" not in m:
                continue
            lit = _folded(m["expr"])
            if lit is not None:
                assert lit == ("lit", "bool", True), (case["name"], m["expr"])
                done += 1
    assert done >= 40, done


def test_nothing_that_reads_the_request_is_folded():
    for expr in ("R.attr.a == 1", "P.id", "now()", "request.resource.kind", "[1, 2].map(x, x + R.attr.a)", "V.x", "timestamp('2021-01-01T00:00:00Z')"):
        assert F.fold(parser.parse(expr)) == parser.parse(expr), expr
    # ... while the constant parts of such an expression are
    got = F.fold(parser.parse('R.attr.tag in "a,b,c".split(",") && P.attr.n > [3, 1, 2].sort()[2]'))
    assert got == parser.parse('R.attr.tag in ["a", "b", "c"] && P.attr.n > 3')


def test_values_without_a_constant_form_give_way_to_the_expression():
    # a hierarchy, an address, bytes, an optional: computed while folding upward, never left in the tree
    for expr in ('hierarchy("a.b.c")', 'ip("10.0.0.1")', 'bytes("x")', 'optional.of(1)', '{1: 2}'):
        assert F.fold(parser.parse(expr)) == parser.parse(expr), expr
    assert F.fold(parser.parse('hierarchy("a.b.c").size()')) == ("lit", "int", 3)
    assert F.fold(parser.parse('cidr("10.0.0.0/8").containsIP("10.1.2.3")')) == ("lit", "bool", True)
    assert F.fold(parser.parse('{1: 2}[1]')) == ("lit", "int", 2)


def test_errors_are_left_to_the_device():
    for expr in ("1 / 0", "[1, 2][5]", '"a" + 1', "9223372036854775807 + 1", '{"a": 1}.b', '"abc".substring(2, 1)', "[1, 'a'].sort()",
                 "1u - 2u", "5 % 0", 'int("x")', 'ip("999.1.1.1")', "int(double(-9223372036854775807 - 1))", "int(9223372036854775808.0)",
                 'basePath("C:x")', 'pathMatch("a", "[")'):
        assert _folded(expr) is None, expr                    # not turned into a value ...
        with pytest.raises(F.FoldError):                      # ... because evaluating it is a CEL error
            F._Eval().ev(parser.parse(expr), {})
        with pytest.raises(celeval.CelError):
            celeval.evaluate(expr, celeval.Env({}, NOW))


STRS = ["", "a", "abc", "a.b.c", "a,b,,c", "  pad  ", "Ünï", "x.y", "ABC", "a.b", "/a/b", "/a/b/c.txt", "/a/*.txt", "../x", "C:\\\\d\\\\e.f",
        "\\\\\\\\h\\\\s\\\\x", "/a/[b"]


def _gen(rng, depth, want):   # noqa: C901 - an expression of (roughly) the wanted type
    def lit_int():
        return str(rng.choice([0, 1, 2, 3, 7, -1, -5, 42, 9223372036854775807]))

    def lit_dbl():
        return rng.choice(["0.0", "1.5", "2.0", "-3.25", "1e3"])

    def lit_str():
        return '"%s"' % rng.choice(STRS)
    if depth <= 0:
        return {"int": lit_int, "dbl": lit_dbl, "str": lit_str, "bool": lambda: rng.choice(["true", "false"]),
                "ilist": lambda: "[%s]" % ", ".join(lit_int() for _ in range(rng.randrange(0, 4))),
                "slist": lambda: "[%s]" % ", ".join(lit_str() for _ in range(rng.randrange(0, 4)))}[want]()
    d = depth - 1
    g = lambda w: _gen(rng, d, w)   # noqa: E731
    if want == "int":
        return rng.choice([
            lambda: "(%s %s %s)" % (g("int"), rng.choice("+-*/%"), g("int")),
            lambda: "size(%s)" % g(rng.choice(["str", "ilist", "slist"])),
            lambda: "%s.indexOf(%s)" % (g("str"), g("str")),
            lambda: "%s.lastIndexOf(%s)" % (g("str"), g("str")),
            lambda: "%s[%s]" % (g("ilist"), rng.choice(["0", "1", "2"])),
            lambda: "hierarchy(%s).size()" % g("str"),
            lambda: "(%s ? %s : %s)" % (g("bool"), g("int"), g("int")),
            lambda: "int(%s)" % g(rng.choice(["dbl", "int"])),
            lit_int])()
    if want == "dbl":
        return rng.choice([lambda: "(%s %s %s)" % (g("dbl"), rng.choice("+-*/"), g("dbl")), lambda: "double(%s)" % g("int"), lit_dbl])()
    if want == "str":
        return rng.choice([
            lambda: "(%s + %s)" % (g("str"), g("str")),
            lambda: "%s.%s()" % (g("str"), rng.choice(["lowerAscii", "upperAscii", "trim", "reverse"])),
            lambda: "%s.replace(%s, %s)" % (g("str"), g("str"), g("str")),
            lambda: "%s.substring(%s)" % (g("str"), rng.choice(["0", "1", "2"])),
            lambda: "%s.charAt(%s)" % (g("str"), rng.choice(["0", "1", "3"])),
            lambda: "%s.join(%s)" % (g("slist"), g("str")),
            lambda: "%s[%s]" % (g("slist"), rng.choice(["0", "1"])),
            lambda: "hierarchy(%s)[%s]" % (g("str"), rng.choice(["0", "1"])),
            lambda: "string(%s)" % g("int"),
            lambda: "%s(%s)" % (rng.choice(["basePath", "dirPath", "extPath", "volumeName"]), g("str")),
            lambda: "joinPath(%s)" % g("slist"),
            lambda: "relPath(%s, %s)" % (g("str"), g("str")),
            lit_str])()
    if want == "ilist":
        return rng.choice([
            lambda: "(%s + %s)" % (g("ilist"), g("ilist")),
            lambda: "%s.%s()" % (g("ilist"), rng.choice(["sort", "distinct", "reverse"])),
            lambda: "%s.filter(x, x %s %s)" % (g("ilist"), rng.choice(["<", ">=", "!="]), g("int")),
            lambda: "%s.map(x, x * 2)" % g("ilist"),
            lambda: "%s(%s, %s)" % (rng.choice(["intersect", "except"]), g("ilist"), g("ilist")),
            lambda: "%s.transformList(i, v, i + v)" % g("ilist"),
            lambda: "lists.range(%s)" % rng.choice(["0", "3", "5"]),
            lambda: "[%s, %s]" % (g("int"), g("int"))])()
    if want == "slist":
        return rng.choice([
            lambda: "%s.split(%s)" % (g("str"), rng.choice(['","', '"."', '"b"'])),
            lambda: "%s.%s()" % (g("slist"), rng.choice(["sort", "distinct", "reverse"])),
            lambda: "%s.map(s, s.upperAscii())" % g("slist"),
            lambda: "%s.filter(s, s.startsWith(%s))" % (g("slist"), g("str")),
            lambda: "[%s, %s]" % (g("str"), g("str"))])()
    return rng.choice([   # bool
        lambda: "(%s %s %s)" % (g("int"), rng.choice(["==", "!=", "<", "<=", ">", ">="]), g(rng.choice(["int", "dbl"]))),
        lambda: "(%s %s %s)" % (g("str"), rng.choice(["==", "!=", "<", ">="]), g("str")),
        lambda: "(%s %s %s)" % (g("bool"), rng.choice(["&&", "||"]), g("bool")),
        lambda: "!%s" % g("bool"),
        lambda: "(%s in %s)" % (g("int"), g("ilist")),
        lambda: "(%s in %s)" % (g("str"), g("slist")),
        lambda: "(%s == %s)" % (g("ilist"), g("ilist")),
        lambda: "%s.%s(%s)" % (g("str"), rng.choice(["startsWith", "endsWith", "contains"]), g("str")),
        lambda: "%s.%s(x, x > %s)" % (g("ilist"), rng.choice(["all", "exists", "exists_one"]), g("int")),
        lambda: "%s(%s, %s)" % (rng.choice(["hasIntersection", "isSubset"]), g("ilist"), g("ilist")),
        lambda: "hierarchy(%s).%s(hierarchy(%s))" % (g("str"), rng.choice(["ancestorOf", "descendentOf", "siblingOf", "overlaps",
                                                                           "immediateParentOf", "immediateChildOf"]), g("str")),
        lambda: "(hierarchy(%s).commonAncestors(hierarchy(%s)) == hierarchy(%s))" % (g("str"), g("str"), g("str")),
        lambda: "sets.%s(%s, %s)" % (rng.choice(["contains", "intersects", "equivalent"]), g("ilist"), g("ilist")),
        lambda: "%s.%s(%s)" % (g("str"), rng.choice(["pathHasPrefix", "pathMatch"]), g("str")),
        lambda: "pathMatchAnyOf(%s, %s)" % (g("str"), g("slist")),
        lambda: rng.choice(["true", "false"])])()


def _same(a, b):
    if isinstance(a, float) and isinstance(b, float):
        return (math.isnan(a) and math.isnan(b)) or a == b
    if isinstance(a, list):
        return isinstance(b, list) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return type(a) is type(b) and a == b or (isinstance(a, int) and isinstance(b, int) and not isinstance(a, bool) and not isinstance(b, bool)
                                             and int(a) == int(b) and isinstance(a, celeval.UInt) == isinstance(b, celeval.UInt))


@pytest.mark.parametrize("seed", range(8))
def test_generated_constant_expressions_against_the_oracle(seed):
    rng = random.Random(9000 + seed)
    folded = errors = 0
    for _ in range(700):
        expr = _gen(rng, rng.randrange(1, 4), rng.choice(["bool", "int", "str", "ilist", "slist", "bool", "dbl"]))
        lit = _folded(expr)
        try:
            want = celeval.evaluate(expr, celeval.Env({}, NOW))
        except celeval.CelError:
            want = celeval.CelError
            errors += 1
        if want is celeval.CelError:
            assert lit is None, (expr, lit)
            continue
        if lit is not None:
            assert _same(_lit_value(lit), want), (expr, lit, want)
            folded += 1
    assert folded > 300 and errors > 20, (folded, errors)


def _math_arg(rng, kind):
    if kind == "int":
        return str(rng.choice([0, 1, -1, 2, -2, 3, 5, -7, 63, 64, 1024, -1024, (1 << 62), -(1 << 62), (1 << 63) - 1, -(1 << 63) + 1]))
    if kind == "uint":
        return "%du" % rng.choice([0, 1, 2, 3, 5, 200, 1024, (1 << 63), (1 << 64) - 1])
    return rng.choice(["0.0", "1.2", "-1.2", "1.5", "-1.5", "2.5", "-0.5", "1e300", "-1e300", "(0.0/0.0)", "(1.0/0.0)", "(-1.0/0.0)", "81.0", "-4.0"])


@pytest.mark.parametrize("seed", range(4))
def test_math_library_on_constants_against_the_oracle(seed):
    """cel-go ext.Math() (conditions.adoc:456-472 lists the functions): the product's folder (cel/fold.py _math) and the oracle
    (celeval._NAMESPACE_FUNCS) are two restatements - the same value, the same type, an error in one is an error in the other."""
    rng = random.Random(31_000 + seed)
    one = ["abs", "ceil", "floor", "round", "trunc", "isNaN", "isInf", "isFinite", "sign", "sqrt", "bitNot"]
    two = ["bitAnd", "bitOr", "bitXor", "bitShiftLeft", "bitShiftRight"]
    folded = errors = 0
    for _ in range(1500):
        if rng.random() < 0.5:
            expr = "math.%s(%s)" % (rng.choice(one), _math_arg(rng, rng.choice(["int", "uint", "dbl"])))
        else:
            expr = "math.%s(%s, %s)" % (rng.choice(two), _math_arg(rng, rng.choice(["int", "uint", "dbl"])), _math_arg(rng, rng.choice(["int", "int", "uint"])))
        lit = _folded(expr)
        try:
            want = celeval.evaluate(expr, celeval.Env({}, NOW))
        except celeval.CelError:
            assert lit is None, (expr, lit)
            errors += 1
            continue
        assert lit is not None, expr
        assert _same(_lit_value(lit), want), (expr, lit, want)
        folded += 1
    assert folded > 600 and errors > 100, (folded, errors)


def test_go_regexp_and_round_vectors():
    """Hand-computed against Go's regexp / math (the folder and the oracle used to share Python's reading of these):
    an empty match abutting the previous match is dropped by ReplaceAllString / FindAll (regexp.go allMatches); RE2's
    \\d \\w \\s are ASCII classes and \\s has no \\v; math.Round is exact (floor(|d| + 0.5) is not)."""
    kats = [
        ('regex.replace("abxd", "x*", "-")', "-a-b-d-"),
        ('regex.extractAll("abxd", "x*")', ["", "", "x", ""]),
        ('regex.replace("٣x", "\\\\d", "N")', "٣x"),
        ('regex.replace("a\\u000bb", "\\\\s", "_")', "a\x0bb"),
        ('regex.replace("a b\\tc", "\\\\s", "_")', "a_b_c"),
        ('regex.replace("banana", "a", "o", 2)', "bonona"),
        ("math.round(0.49999999999999994)", 0.0),
        ("math.round(4503599627370497.0)", 4503599627370497.0),
        ("math.round(-2.5)", -3.0),
        ("math.round(2.5)", 3.0),
    ]
    for expr, want in kats:
        assert celeval.evaluate(expr, celeval.Env({}, NOW)) == want, expr
        lit = _folded(expr)
        assert lit is not None and _lit_value(lit) == want, (expr, lit)
