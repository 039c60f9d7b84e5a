"""Mines the Go-coded known-answer tests of the Cerbos CEL library - TestCerbosLib (internal/conditions/cerbos_lib_test.go:26-193):
each expression must evaluate to true, or fail where wantErr is set - into tests/golden/cerbos_lib_kats.json.
   python tools/make_golden_cerbos_lib.py        (needs /root/reference; the fixture travels, the reference does not)"""
import json
import os
import re

REF = "/root/reference/internal/conditions/cerbos_lib_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cerbos_lib_kats.json")

src = open(REF).read()
body = src[src.index("func TestCerbosLib("):src.index("env, err := cel.NewEnv(conditions.CerbosCELLib())")]
cases = []
for m in re.finditer(r"\{expr: `([^`]*)`(, wantErr: true)?\}", body):
    cases.append({"expr": m.group(1), "wantErr": bool(m.group(2))})
assert len(cases) > 140, len(cases)
json.dump({"source": "internal/conditions/cerbos_lib_test.go:26-193 (TestCerbosLib)", "cases": cases}, open(OUT, "w"), indent=1)
print("wrote", OUT, len(cases), "cases,", sum(c["wantErr"] for c in cases), "expecting an error")
