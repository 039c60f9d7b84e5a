"""Mines the Go-coded known-answer tables of the reference's naming rules - internal/namer/namer_test.go (TestFQN, TestFQNTree,
TestFQNSpecialChars with the xxhash64 module ids) and internal/conditions/identifiers_test.go (TestValidateIdentifiers) - into
tests/golden/namer_vectors.json.  The policies TestFQN / TestFQNTree name come from internal/test/policy.go (GenResourcePolicy ...:
resource leave_request, principal donald_duck, derived roles my_derived_roles, exports my_constants / my_variables, version default).
   python tools/make_golden_namer.py        (needs /root/reference; the fixture travels, the reference does not)"""
import json
import os
import re

REF = "/root/reference/internal"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "namer_vectors.json")

src = open(os.path.join(REF, "namer/namer_test.go")).read()


def func_body(name):
    a = src.index("func %s(" % name)
    b = src.find("\nfunc ", a + 1)
    return src[a:b if b >= 0 else len(src)]


_GEN = {"GenDerivedRoles": "derivedRoles", "GenExportConstants": "exportConstants", "GenExportVariables": "exportVariables",
        "GenResourcePolicy": "resourcePolicy", "GenPrincipalPolicy": "principalPolicy"}
# internal/test/policy.go: the names the generators give
mk = open(os.path.join(REF, "test/policy.go")).read()
for needle in ('"leave_request"', '"donald_duck"', '"my_derived_roles"', '"my_constants"', '"my_variables"'):
    assert needle in mk, needle
_NAMES = {"derivedRoles": ("name", "my_derived_roles"), "exportConstants": ("name", "my_constants"), "exportVariables": ("name", "my_variables"),
          "resourcePolicy": ("resource", "leave_request"), "principalPolicy": ("principal", "donald_duck")}


def policy_cases(name, want_re):
    out = []
    body = func_body(name)
    body = body[:body.index("for _, tc := range testCases")]
    for entry in re.split(r"\n\t\t\{\n", body)[1:]:   # one `{ name: ..., policy: ..., want: ... }` each
        title = re.search(r'name:\s*"([^"]+)"', entry).group(1)
        gen = re.search(r"test\.(Gen\w+)\(", entry).group(1)
        kind = _GEN[gen]
        key, val = _NAMES[kind]
        pol = {key: val}
        if kind in ("resourcePolicy", "principalPolicy"):
            pol["version"] = "default"
            sc = re.search(r'\.Scope = "([^"]*)"', entry)
            if sc:
                pol["scope"] = sc.group(1)
        want = re.search(r"want:\s*" + want_re, entry, re.S).group(1)
        out.append({"name": title, "policy": {"apiVersion": "api.cerbos.dev/v1", kind: pol}, "want": want})
    return out


fqn = policy_cases("TestFQN", r'"([^"]+)"')
tree = policy_cases("TestFQNTree", r"\[\]string\{(.*?)\}")
for c in tree:
    c["want"] = re.findall(r'"([^"]+)"', c["want"])
special = [{"policyName": a, "kind": "resource" if "Resource" in b else "principal", "version": "default", "scope": "a.b.c", "wantFQN": c, "wantModuleID": d}
           for a, b, c, d in re.findall(r'policyName:\s*"([^"]+)",\s*fqnFunc:\s*namer\.(\w+),\s*wantFQN:\s*"([^"]+)",\s*wantModuleID:\s*"(\d+)"',
                                        func_body("TestFQNSpecialChars"))]
ids = open(os.path.join(REF, "conditions/identifiers_test.go")).read()
valid = re.findall(r'"([^"]*)"', ids[ids.index("valid := []string{"):ids.index("for _, identifier := range valid")])
invalid = re.findall(r'"([^"]*)"', ids[ids.index("invalid := []string{"):ids.index("for _, identifier := range invalid")])
assert len(fqn) == 7 and len(tree) == 7 and len(special) == 8 and len(valid) == 11 and len(invalid) == 8, (len(fqn), len(tree), len(special), len(valid), len(invalid))
json.dump({"source": "internal/namer/namer_test.go:17-143,223-288; internal/conditions/identifiers_test.go:13-46",
           "fqn": fqn, "fqn_tree": tree, "special_chars": special, "identifiers": {"valid": valid, "invalid": invalid}}, open(OUT, "w"), indent=1)
print("wrote", OUT)
