"""The reference's file-path helpers (internal/conditions/crosspath): its own known-answer tables
(crosspath_test.go, mined into tests/golden/crosspath_vectors.json by tools/make_golden_crosspath.py) against

* oracle/crosspath.py - the checker's restatement, and
* cerbos_amd/cel/crosspath.py - the product's (what the lowering folds constant path expressions with),

written apart from each other and then held against each other on generated paths."""
import random

import pytest

from cerbos_amd.cel import crosspath as product
from helpers import load_json
from oracle import crosspath as oracle

V = load_json("crosspath_vectors.json")
BOTH = [pytest.param(oracle, id="oracle"), pytest.param(product, id="product")]


def cases(key):
    return V[key]["cases"]


@pytest.mark.parametrize("m", BOTH)
def test_round_trip(m):
    for p in cases("round_trip"):
        assert m.decode(m.encode(p)) == p


@pytest.mark.parametrize("m", BOTH)
def test_base_ext_volume(m):
    for c in cases("base"):
        assert m.base(c["path"]) == c["want"], c
    for c in cases("ext"):
        assert m.ext(c["path"]) == c["want"], c
    for c in cases("volume_name"):
        assert m.volume_name(c["path"]) == c["want"], c


@pytest.mark.parametrize("m", BOTH)
def test_dir(m):
    for c in cases("dir"):
        if c["expectErr"]:
            with pytest.raises(m.PathError):
                m.dir_(c["path"])
        else:
            assert m.dir_(c["path"]) == c["want"], c


@pytest.mark.parametrize("m", BOTH)
def test_join_match_rel(m):
    for c in cases("join"):
        assert m.join(c["paths"]) == c["want"], c
    for c in cases("match"):
        assert m.match(*c["paths"]) is c["want"], c
    for c in cases("rel"):
        assert m.rel(*c["paths"]) == c["want"], c


# Go's documented examples of path/filepath (GOOS=linux): Clean's rules, Match's grammar, Rel's errors
@pytest.mark.parametrize("m", BOTH)
def test_go_filepath_documented_behaviour(m):
    for p, want in (("", "."), ("a/c", "a/c"), ("a//c", "a/c"), ("a/c/.", "a/c"), ("a/c/b/..", "a/c"), ("/../a/c", "/a/c"),
                    ("/../a/b/../././/c", "/a/c"), ("../../a", "../../a"), ("a/../..", ".."), ("/", "/"), ("//", "/"), ("a/..", "."),
                    ("abc/def/../../..", ".."), ("/abc/def/../../..", "/")):
        assert m.fp_clean(p) == want, p
    for pat, name, want in (("abc", "abc", True), ("*", "abc", True), ("*c", "abc", True), ("a*", "a", True), ("a*", "ab/c", False),
                            ("a*/b", "abc/b", True), ("a*/b", "a/c/b", False), ("a*b*c*d*e*/f", "axbxcxdxe/f", True),
                            ("a*b*c*d*e*/f", "axbxcxdxexxx/fff", False), ("a*b?c*x", "abxbbxdbxebxczzx", True),
                            ("a*b?c*x", "abxbbxdbxebxczzy", False), ("ab[c]", "abc", True), ("ab[b-d]", "abc", True),
                            ("ab[e-g]", "abc", False), ("ab[^c]", "abc", False), ("ab[^b-d]", "abc", False), ("ab[^e-g]", "abc", True),
                            ("a\\*b", "a*b", True), ("a\\*b", "ab", False), ("a?b", "a/b", False), ("a*b", "a/b", False),
                            ("[\\]a]", "]", True), ("[\\-]", "-", True), ("[x\\-]", "x", True), ("[x\\-]", "z", False),
                            ("[a-b-c]", "a", None), ("[", "a", None), ("[^", "a", None), ("[^bc", "a", None), ("a[", "a", None),
                            ("a[", "ab", None), ("a[", "x", None), ("a/b[", "x", None), ("[]a]", "]", None), ("[-]", "-", None),
                            ("[x-]", "x", None), ("[-x]", "x", None), ("\\", "a", None), ("*x", "xxx", True), ("", "", True), ("", "a", False)):
        if want is None:
            with pytest.raises(m.PathError):
                m.fp_match(pat, name)
        else:
            assert m.fp_match(pat, name) is want, (pat, name)
    for b, t, want in (("a/b", "a/b", "."), ("a/b/.", "a/b", "."), ("a/b", "a/b/c", "c"), ("a/b", "a/b/../c", "../c"), ("a/b/c", "a/c/d", "../../c/d"),
                       ("a/b", "c/d", "../../c/d"), ("../../a/b", "../../a/b/c/d", "c/d"), ("/a/b", "/a/b/../c", "../c"), ("/a/b/c", "/a/c/d", "../../c/d"),
                       (".", "a/b", "a/b"), (".", "..", ".."), ("/", "/a/b", "a/b"), ("/ab/cd", "/ab/c", "../c"), ("..", ".", None),
                       ("..", "a", None), ("../..", "..", None), ("a", "/a", None), ("/a", "a", None)):
        if want is None:
            with pytest.raises(m.PathError):
                m.fp_rel(b, t)
        else:
            assert m.fp_rel(b, t) == want, (b, t)
    assert [m.fp_base(p) for p in ("", ".", "/.", "/", "////", "x/", "abc", "abc/def", "a/b/.x", "a/b/c.", "a/b/c.x")] == \
        [".", ".", ".", "/", "/", "x", "abc", "def", ".x", "c.", "c.x"]
    assert [m.fp_dir(p) for p in ("", ".", "/.", "/", "/foo", "x/", "abc", "abc/def", "a/b/.x", "a/b/c.", "a/b/c.x")] == \
        [".", ".", "/", "/", "/", "x", ".", "abc", "a/b", "a/b", "a/b"]
    assert [m.fp_ext(p) for p in ("path.go", "path.pb.go", "a.dir/b", "a.dir/b.go", "a.dir/")] == [".go", ".go", "", ".go", ""]
    assert m.fp_join("a", "b") == "a/b" and m.fp_join("a", "") == "a" and m.fp_join("", "b") == "b" and m.fp_join("/", "a") == "/a" \
        and m.fp_join("a/", "b") == "a/b" and m.fp_join("", "") == "" and m.fp_join("a", "../..", "b") == "../b"


def _random_path(rng):
    style = rng.choice(("unix", "unc", "drive", "win", "odd"))
    parts = [rng.choice(("a", "bb", "c.txt", "..", ".", "", "d.e.f", "*", "?x", "[a-c]", "é", "x y")) for _ in range(rng.randrange(0, 5))]
    if style == "unix":
        return rng.choice(("", "/", "./", "../")) + "/".join(parts)
    if style == "unc":
        return "\\\\" + "\\".join(["host", "share"][:rng.randrange(0, 3)] + parts)
    if style == "drive":
        return rng.choice(("C:", "z:", "C:\\", "C:/", "C:x")) + "\\".join(parts)
    if style == "win":
        return rng.choice(("", "\\", "..\\")) + "\\".join(parts)
    return "".join(rng.choice("/\\.:ab*[]?-^") for _ in range(rng.randrange(0, 8)))


def _outcome(fn, *a):
    try:
        return fn(*a)
    except Exception as x:   # noqa: BLE001 - the two modules have their own error classes
        assert type(x).__name__ in ("PathError", "BadPattern"), (fn, a, x)
        return "error"


def test_product_and_oracle_agree_on_generated_paths():
    rng = random.Random(20260922)
    for _ in range(20000):
        p, q = _random_path(rng), _random_path(rng)
        for name in ("base", "dir_", "ext", "volume_name"):
            assert _outcome(getattr(product, name), p) == _outcome(getattr(oracle, name), p), (name, p)
        for name in ("match", "rel", "has_prefix"):
            assert _outcome(getattr(product, name), p, q) == _outcome(getattr(oracle, name), p, q), (name, p, q)
        ps = [p, q, _random_path(rng)][:rng.randrange(0, 4)]
        assert _outcome(product.join, ps) == _outcome(oracle.join, ps), ps
        assert _outcome(product.match_any_of, p, ps) == _outcome(oracle.match_any_of, p, ps), (p, ps)
