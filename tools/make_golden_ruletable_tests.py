#!/usr/bin/env python3
"""Mines the known-answer tests the reference keeps in Go test code on the hot path itself:

  internal/ruletable/cel_errors_test.go          TestCELErrorsCheck        (fail-open semantics of CEL runtime errors)
  internal/ruletable/strict_evaluation_test.go   TestStrictEvaluationCheck (EvalParams.StrictEvaluation: an error denies)
  ... and their planner halves, TestCELErrorsPlan / TestStrictEvaluationPlan (the filter's kind and the reported expressions)

Both run RuleTable.Check over six small policies built in newCELErrorsHarness and assert, per case, the effect of every
action (sometimes the policy) and the ORDERED list of expressions in CheckOutput.EvaluationErrors.  The policies are Go
struct literals of a regular shape and the cases are t.Run blocks of two shapes; both are read with regular
expressions and written as tests/golden/ruletable_cel_errors.json.  Run in the build container (needs /root/reference):

    python tools/make_golden_ruletable_tests.py
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/internal/ruletable"
OUT = os.path.join(ROOT, "tests", "golden", "ruletable_cel_errors.json")
API = "api.cerbos.dev/v1"


def consts(src):
    out = {}
    for name, lit in re.findall(r'^\s*(\w+)\s*=\s*(`[^`]*`|"(?:[^"\\]|\\.)*")\s*$', src, re.M):
        out[name] = go_string(lit)
    return out


def go_string(lit):
    if lit.startswith("`"):
        return lit[1:-1]
    return json.loads(lit)


def expr_of(tok, k):
    tok = tok.strip()
    return go_string(tok) if tok[0] in "`\"" else k[tok]


def strings(body):
    return [go_string(x) for x in re.findall(r'`[^`]*`|"(?:[^"\\]|\\.)*"', body)]


def policies(src, k):
    docs = []
    for m in re.finditer(r"(\w+) := &policyv1\.Policy\{(.*?)\n\t\}\n", src, re.S):
        body = m.group(2)
        if "Policy_DerivedRoles" in body:
            name = re.search(r'Name:\s*("[^"]+"),\n\s*(?:Variables|Definitions)', body).group(1)
            dr = {"name": go_string(name), "definitions": []}
            v = re.search(r"Variables: &policyv1\.Variables\{Local: map\[string\]string\{(.*?)\}\}", body)
            if v:
                dr["variables"] = {"local": {go_string(a): expr_of(b, k) for a, b in re.findall(r'("[^"]+"):\s*([^,}]+)', v.group(1))}}
            for d in re.finditer(r'\{Name: ("[^"]+"), ParentRoles: \[\]string\{([^}]*)\}(?:, Condition: cond\(([^)]+)\))?\}', body):
                e = {"name": go_string(d.group(1)), "parentRoles": strings(d.group(2))}
                if d.group(3):
                    e["condition"] = {"match": {"expr": expr_of(d.group(3), k)}}
                dr["definitions"].append(e)
            docs.append({"apiVersion": API, "derivedRoles": dr})
        else:
            rp = {"resource": go_string(re.search(r'Resource:\s*("[^"]+")', body).group(1)),
                  "version": go_string(re.search(r'(?<![A-Za-z])Version:\s*("[^"]+")', body).group(1)), "rules": []}
            imp = re.search(r"ImportDerivedRoles: \[\]string\{([^}]*)\}", body)
            if imp:
                rp["importDerivedRoles"] = strings(imp.group(1))
            v = re.search(r"Variables:\s*&policyv1\.Variables\{Local: map\[string\]string\{(.*?)\}\}", body)
            if v:
                rp["variables"] = {"local": {go_string(a): expr_of(b, k) for a, b in re.findall(r'("[^"]+"):\s*([^,}]+)', v.group(1))}}
            for r in re.finditer(r"\{Actions: \[\]string\{([^}]*)\}, (Roles|DerivedRoles): \[\]string\{([^}]*)\}, Effect: effectv1\.Effect_(\w+)(?:, Condition: cond\(([^)]+)\))?\}", body):
                rule = {"actions": strings(r.group(1)), "effect": r.group(4)}
                rule["roles" if r.group(2) == "Roles" else "derivedRoles"] = strings(r.group(3))
                if r.group(5):
                    rule["condition"] = {"match": {"expr": expr_of(r.group(5), k)}}
                rp["rules"].append(rule)
            docs.append({"apiVersion": API, "resourcePolicy": rp})
    return docs


def amount_of(tok, local):
    tok = tok.strip()
    tok = local.get(tok, tok)
    if tok == "nil":
        return {"absent": True}
    m = re.match(r"structpb\.NewNumberValue\(([^)]+)\)", tok)
    if m:
        return {"number": float(m.group(1))}
    m = re.match(r'structpb\.NewStringValue\(("[^"]*")\)', tok)
    if m:
        return {"string": go_string(m.group(1))}
    raise ValueError(tok)


def cases(src, func, strict, k):
    body = src[src.index("func %s(" % func):]
    body = body[:body.index("\n}\n") + 3]
    local = dict(re.findall(r"(\w+) := (structpb\.New\w+Value\([^)]*\))", body))
    out = []
    for m in re.finditer(r't\.Run\("(\w+)", func\(t \*testing\.T\) \{(.*?)\n\t\}\)', body, re.S):
        name, blk = m.group(1), m.group(2)
        errs = re.search(r"assertCELErrors\(t, (?:entries|out\.EvaluationErrors)((?:, [^,)]+)*)\)", blk)
        want_errs = [expr_of(x, k) for x in errs.group(1).split(",")[1:]] if errs else None
        c = re.search(r'h\.check\(t, ([^,]+(?:\([^)]*\))?), ("[^"]+"), ("[^"]+"), (.+?)\)\n', blk)
        if c:
            case = {"name": name, "strict": strict, "kind": go_string(c.group(2)), "actions": [go_string(c.group(3))],
                    "amount": amount_of(c.group(4), local), "wantErrorExpressions": want_errs, "wantEffects": {}, "wantPolicies": {}}
            e = re.search(r"require\.Equal\(t, effectv1\.Effect_(\w+), effect\)", blk)
            if e:
                case["wantEffects"][case["actions"][0]] = e.group(1)
        else:
            c = re.search(r'checkInputActions\(("[^"]+"), ([^,]+(?:\([^)]*\))?), ((?:"[^"]+"(?:, )?)+)\)', blk)
            case = {"name": name, "strict": strict, "kind": go_string(c.group(1)), "actions": strings(c.group(3)),
                    "amount": amount_of(c.group(2), local), "wantErrorExpressions": want_errs, "wantEffects": {}, "wantPolicies": {}}
            for eff, act in re.findall(r'require\.Equal\(t, effectv1\.Effect_(\w+), out\.Actions\[("[^"]+")\]\.GetEffect\(\)\)', blk):
                case["wantEffects"][go_string(act)] = eff
            for pol, act in re.findall(r'require\.Equal\(t, ("[^"]+"), out\.Actions\[("[^"]+")\]\.GetPolicy\(\)\)', blk):
                case["wantPolicies"][go_string(act)] = go_string(pol)
        out.append(case)
    return out


def plan_cases(src, func, strict, k):
    """TestCELErrorsPlan / TestStrictEvaluationPlan: h.plan(t, params, kind, action, amount) -> the filter's kind and the expressions of
    PlanResourcesOutput.EvaluationErrors."""
    body = src[src.index("func %s(" % func):]
    body = body[:body.index("\n}\n") + 3]
    local = dict(re.findall(r"(\w+) := (structpb\.New\w+Value\([^)]*\))", body))
    out = []
    for m in re.finditer(r't\.Run\("(\w+)", func\(t \*testing\.T\) \{(.*?)\n\t\}\)', body, re.S):
        name, blk = m.group(1), m.group(2)
        c = re.search(r'h\.plan\(t, ([^,]+(?:\([^)]*\))?), ("[^"]+"), ("[^"]+"), (.+?)\)\n', blk)
        kind = re.search(r"require\.Equal\(t, enginev1\.PlanResourcesFilter_(KIND_\w+), kind\)", blk)
        errs = re.search(r"assertCELErrors\(t, entries((?:, [^,)]+)*)\)", blk)
        out.append({"name": name, "strict": strict, "kind": go_string(c.group(2)), "action": go_string(c.group(3)),
                    "amount": amount_of(c.group(4), local), "wantKind": kind.group(1),
                    "wantErrorExpressions": [expr_of(x, k) for x in errs.group(1).split(",")[1:]]})
    return out


def main():
    a = open(os.path.join(REF, "cel_errors_test.go"), encoding="utf-8").read()
    b = open(os.path.join(REF, "strict_evaluation_test.go"), encoding="utf-8").read()
    k = consts(a)
    doc = {"source": ["internal/ruletable/cel_errors_test.go: newCELErrorsHarness, TestCELErrorsCheck",
                      "internal/ruletable/strict_evaluation_test.go: TestStrictEvaluationCheck"],
           "principal": {"id": "sam", "roles": ["user"]}, "resourceId": "1", "requestId": "1",
           "policies": policies(a, k),
           "cases": cases(a, "TestCELErrorsCheck", False, k) + cases(b, "TestStrictEvaluationCheck", True, k),
           # ... and the same harness through RuleTable.Plan (TestCELErrorsPlan, TestStrictEvaluationPlan)
           "planCases": plan_cases(a, "TestCELErrorsPlan", False, k) + plan_cases(b, "TestStrictEvaluationPlan", True, k)}
    assert len(doc["policies"]) == 6 and len(doc["cases"]) >= 22 and len(doc["planCases"]) >= 17, (len(doc["policies"]), len(doc["cases"]), len(doc["planCases"]))
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(doc, f, sort_keys=True, indent=1, ensure_ascii=False)
        f.write("\n")
    print("wrote", OUT, len(doc["policies"]), "policies,", len(doc["cases"]), "cases,", len(doc["planCases"]), "plan cases")


if __name__ == "__main__":
    sys.exit(main())
