"""Mines the reference's policy-compiler test cases - internal/test/testdata/compile/*.yaml with their `.input` archives (txtar:
the policy files of the compilation unit) and `.golden` results (the RunnablePolicySet as protojson), the table
internal/compile/compile_test.go:40-88 (TestCompile) runs - into tests/golden/compile_cases.json.

Kept per case: the main definition, the policy files as TEXT (positions in the expected errors are lines / columns of it), the
expected errors (file, error kind, description, position) or the golden policy set without the type-checked expression trees
(`checked`: cel-go's CheckedExpr, which nothing on the hot path reads - the rule table keeps `original` and re-compiles).
   python tools/make_golden_compile.py        (needs /root/reference; the fixture travels, the reference does not)"""
import json
import os

import yaml

REF = "/root/reference/internal/test/testdata/compile"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "compile_cases.json")


def txtar(text):
    """golang.org/x/tools/txtar: `-- name --` lines start a file; what precedes the first is a comment."""
    files, name, lines = {}, None, []
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("-- ") and s.endswith(" --") and len(s) >= 7:
            if name is not None:
                files[name] = "\n".join(lines) + "\n"
            name, lines = s[3:-3].strip(), []
        else:
            lines.append(line)
    if name is not None:
        body = "\n".join(lines)
        files[name] = body if body.endswith("\n") or body == "" else body + "\n"
    return files


def strip_checked(x):
    if isinstance(x, dict):
        return {k: strip_checked(v) for k, v in x.items() if k != "checked"}
    if isinstance(x, list):
        return [strip_checked(v) for v in x]
    return x


cases = []
for fn in sorted(os.listdir(REF)):
    if not fn.endswith(".yaml"):
        continue
    tc = [d for d in yaml.safe_load_all(open(os.path.join(REF, fn))) if d][0]
    case = {"name": fn[:-5], "mainDef": tc["mainDef"], "files": txtar(open(os.path.join(REF, fn + ".input")).read()),
            "wantErrors": tc.get("wantErrors") or [], "wantVariables": tc.get("wantVariables") or []}
    assert case["mainDef"] in case["files"], fn
    golden = os.path.join(REF, fn + ".golden")
    if os.path.exists(golden):
        case["golden"] = strip_checked(json.load(open(golden)))
    assert bool(case["wantErrors"]) != ("golden" in case), fn
    cases.append(case)
assert len(cases) >= 30, len(cases)
json.dump({"source": "internal/test/testdata/compile (internal/compile/compile_test.go:40-88)", "cases": cases}, open(OUT, "w"), indent=1, sort_keys=True)
print("wrote", OUT, len(cases), "cases,", sum(bool(c["wantErrors"]) for c in cases), "expecting errors")
