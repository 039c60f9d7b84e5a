"""Mines the Go-coded known-answer tables of the reference's file-path helpers - internal/conditions/crosspath/crosspath_test.go
(TestEncodeAndDecode, TestBase, TestDir, TestExt, TestJoin, TestMatch, TestRel, TestVolumeName) - into
tests/golden/crosspath_vectors.json.
   python tools/make_golden_crosspath.py        (needs /root/reference; the fixture travels, the reference does not)"""
import json
import os
import re

REF = "/root/reference/internal/conditions/crosspath/crosspath_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "crosspath_vectors.json")


def tokens(text):
    """Go source -> tokens: ("str", value) for both literal forms, ("sym", char) and ("id", word); comments dropped."""
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if c == "`":
            j = text.index("`", i + 1)
            yield "str", text[i + 1:j]
            i = j + 1
        elif c == '"':
            j, out = i + 1, []
            while text[j] != '"':
                if text[j] == "\\":
                    out.append({"n": "\n", "t": "\t", "\\": "\\", '"': '"'}[text[j + 1]])
                    j += 2
                else:
                    out.append(text[j])
                    j += 1
            yield "str", "".join(out)
            i = j + 1
        elif text.startswith("//", i):
            i = text.index("\n", i)
        elif c.isalpha() or c == "_":
            j = i
            while text[j].isalnum() or text[j] == "_":
                j += 1
            yield "id", text[i:j]
            i = j
        elif c.isspace():
            i += 1
        else:
            yield "sym", c
            i += 1


def table(src, func):
    """The entries of `testCases := []T{...}` in one test function, as dicts (a bare list of strings: as strings)."""
    body = src[src.index("func %s(" % func):]
    body = body[:body.index("\n}\n") + 3]
    body = body[body.index("testCases :="):body.index("for idx, testCase")]
    toks = list(tokens(body))
    # skip to the brace that opens the table: after `[]string` or after the struct type's closing brace
    k = next(i for i, t in enumerate(toks) if t == ("sym", "["))
    if toks[k + 2] == ("id", "struct"):
        k = next(i for i in range(k, len(toks)) if toks[i] == ("sym", "}")) + 1
    else:
        k += 3
    assert toks[k] == ("sym", "{"), toks[k:k + 3]
    k += 1
    entries = []
    while toks[k] != ("sym", "}"):
        if toks[k][0] == "str":
            entries.append(toks[k][1])
            k += 1
        else:
            assert toks[k] == ("sym", "{")
            k += 1
            e = {}
            while toks[k] != ("sym", "}"):
                key = toks[k][1]
                assert toks[k + 1] == ("sym", ":"), toks[k:k + 3]
                k += 2
                if toks[k][0] == "str":
                    e[key] = toks[k][1]
                    k += 1
                elif toks[k] == ("id", "paths"):
                    k += 2
                    vals = []
                    while toks[k] != ("sym", ")"):
                        if toks[k][0] == "str":
                            vals.append(toks[k][1])
                        k += 1
                    e[key] = vals
                    k += 1
                elif toks[k][0] == "id":
                    e[key] = {"nil": None, "true": True, "false": False}[toks[k][1]]
                    k += 1
                else:
                    raise AssertionError(toks[k:k + 3])
                if toks[k] == ("sym", ","):
                    k += 1
            entries.append(e)
            k += 1
        if toks[k] == ("sym", ","):
            k += 1
    return entries


src = open(REF).read()
lines = src.split("\n")


def where(func):
    a = next(i for i, l in enumerate(lines) if l.startswith("func %s(" % func)) + 1
    b = next(i for i in range(a, len(lines)) if lines[i] == "}") + 1
    return "crosspath_test.go:%d-%d" % (a, b)


out = {"source": "internal/conditions/crosspath/crosspath_test.go"}
for func, key in (("TestEncodeAndDecode", "round_trip"), ("TestBase", "base"), ("TestDir", "dir"), ("TestExt", "ext"), ("TestJoin", "join"),
                  ("TestMatch", "match"), ("TestRel", "rel"), ("TestVolumeName", "volume_name")):
    out[key] = {"source": where(func), "cases": table(src, func)}
for c in out["join"]["cases"]:
    c["paths"] = c["paths"] or []
for c in out["match"]["cases"]:
    c.setdefault("want", False)
for c in out["dir"]["cases"]:
    c.setdefault("expectErr", False)
n = sum(len(v["cases"]) for v in out.values() if isinstance(v, dict))
assert n > 100, n
assert not re.search(r"nolint", json.dumps(out))
json.dump(out, open(OUT, "w"), indent=1)
print("wrote", OUT, n, "vectors")
