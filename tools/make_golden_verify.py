"""Mines the reference's policy-test-framework fixtures for decision vectors (run where /root/reference exists).

internal/test/testdata/verify/cases/case_NNN.yaml{,.input,.golden} (internal/verify/verify_test.go:44-120): each
case is a txtar archive of test suites + principal / resource / auxData fixtures that `verify.Verify` runs against
the engine built from testdata/store, and a golden TestResults JSON that records - per test, principal, resource
and action - the effect the REFERENCE ENGINE returned (details.success.effect, or details.failure.actual).  Those
are reference outputs for this path on inputs the engine cases do not have (test-level `now`, JWT claims,
globals, default versions, lenient scope search), so they are pinned here as tests/golden/verify_vectors.json:
    {"suite", "test", "now", "globals", "lenient", "strict", "defaultPolicyVersion", "defaultScope", "input": CheckInput,
     "want": {action: effect}}
"""
import glob
import json
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("CERBOS_REFERENCE", "/root/reference")
CASES = os.path.join(REF, "internal/test/testdata/verify/cases")
OUT = os.path.join(ROOT, "tests", "golden", "verify_vectors.json")


def txtar(text):
    files, name, buf = {}, None, []
    for line in text.splitlines(keepends=True):
        if line.startswith("-- ") and line.rstrip().endswith(" --"):
            if name is not None:
                files[name] = "".join(buf)
            name, buf = line.strip()[3:-3].strip(), []
        elif name is not None:
            buf.append(line)
    if name is not None:
        files[name] = "".join(buf)
    return files


def load(text):
    return yaml.safe_load(text)      # JSON is YAML; anchors and merge keys resolve here


def fixtures_for(files, suite_path):
    """testdata/{principals,resources,auxdata}.{yaml,yml,json} next to the suite (verify/test_fixture.go)."""
    base = os.path.dirname(suite_path)
    out = {"principals": {}, "resources": {}, "auxData": {}, "principalGroups": {}, "resourceGroups": {}}
    for stem, keys in (("principals", ("principals", "principalGroups")), ("resources", ("resources", "resourceGroups")),
                       ("auxdata", ("auxData",))):
        for ext in ("yaml", "yml", "json"):
            p = os.path.join(base, "testdata", "%s.%s" % (stem, ext)).lstrip("./")
            if p in files:
                try:
                    doc = load(files[p]) or {}
                except yaml.YAMLError:
                    continue
                if not isinstance(doc, dict):
                    continue
                for k in keys:
                    out[k].update(doc.get(k) or {})
    return out


def expand(names, groups, group_names):
    out = list(names or [])
    for g in group_names or []:
        out.extend((groups.get(g) or {}).get("principals") or (groups.get(g) or {}).get("resources") or [])
    return out


def main():
    vectors = []
    for gpath in sorted(glob.glob(os.path.join(CASES, "*.golden"))):
        case = os.path.basename(gpath)[:-len(".yaml.golden")]
        with open(gpath, encoding="utf-8") as fh:
            golden = json.load(fh)
        with open(gpath[:-len(".golden")] + ".input", encoding="utf-8") as fh:
            files = txtar(fh.read())
        for suite_res in golden.get("suites") or []:
            spath = suite_res.get("file", "")
            if spath not in files:
                continue
            try:
                suite = load(files[spath]) or {}
            except yaml.YAMLError:
                continue
            if not isinstance(suite, dict):
                continue
            fx = fixtures_for(files, spath)
            for k in fx:
                fx[k].update(suite.get(k) or {})      # inline fixtures of the suite win
            sopt = suite.get("options") or {}
            tests = {t.get("name"): t for t in suite.get("tests") or [] if isinstance(t, dict)}
            for tc in suite_res.get("testCases") or []:
                t = tests.get(tc.get("name"))
                if t is None:
                    continue
                opt = dict(sopt)
                opt.update(t.get("options") or {})
                tin = t.get("input") or {}
                aux = fx["auxData"].get(tin.get("auxData")) if tin.get("auxData") else None
                for pr in tc.get("principals") or []:
                    p = fx["principals"].get(pr.get("name"))
                    for rr in pr.get("resources") or []:
                        r = fx["resources"].get(rr.get("name"))
                        if p is None or r is None:
                            continue
                        want = {}
                        for a in rr.get("actions") or []:
                            d = a.get("details") or {}
                            eff = (d.get("success") or {}).get("effect") or (d.get("failure") or {}).get("actual")
                            if eff:
                                want[a["name"]] = eff
                        if not want:
                            continue
                        inp = {"requestId": "%s/%s" % (case, tc["name"]), "principal": p, "resource": r, "actions": list(want)}
                        if aux:
                            inp["auxData"] = {k: aux[k] for k in ("jwt", "jwts") if aux.get(k)}
                        vectors.append({"suite": "%s/%s" % (case, spath), "test": tc["name"], "now": opt.get("now"),
                                        "globals": opt.get("globals") or {}, "lenient": bool(opt.get("lenientScopeSearch")), "strict": bool(opt.get("strictEvaluation")),
                                        "defaultPolicyVersion": opt.get("defaultPolicyVersion") or "default",
                                        "defaultScope": opt.get("defaultScope") or "", "input": inp, "want": want})
    # testdata/store/tests/*_test.yaml: the store's own policy tests (inline fixtures, `expected` effects; an action
    # an expectation does not name must be denied - verify/run_test_suite.go)
    for spath in sorted(glob.glob(os.path.join(REF, "internal/test/testdata/store/tests", "*_test.yaml"))):
        with open(spath, encoding="utf-8") as fh:
            suite = load(fh.read()) or {}
        sopt = suite.get("options") or {}
        for t in suite.get("tests") or []:
            opt = dict(sopt)
            opt.update(t.get("options") or {})
            tin = t.get("input") or {}
            for exp in t.get("expected") or []:
                p, r = (suite.get("principals") or {}).get(exp["principal"]), (suite.get("resources") or {}).get(exp["resource"])
                if p is None or r is None:
                    continue
                want = {a: (exp.get("actions") or {}).get(a, "EFFECT_DENY") for a in tin.get("actions") or []}
                inp = {"requestId": "store/tests/%s" % t["name"], "principal": p, "resource": r, "actions": list(want)}
                vectors.append({"suite": "store/tests/%s" % os.path.basename(spath), "test": t["name"], "now": opt.get("now"),
                                "globals": opt.get("globals") or {}, "lenient": bool(opt.get("lenientScopeSearch")),
                                "strict": bool(opt.get("strictEvaluation")),
                                "defaultPolicyVersion": opt.get("defaultPolicyVersion") or "default",
                                "defaultScope": opt.get("defaultScope") or "", "input": inp, "want": want})
    # the framework's own test cases repeat the same suites: keep one of each (input, options, result)
    seen, uniq = set(), []
    for v in vectors:
        key = json.dumps({k: v[k] for k in v if k not in ("suite", "test")} | {"input": {k: x for k, x in v["input"].items() if k != "requestId"}},
                         sort_keys=True, default=str)
        if key not in seen:
            seen.add(key)
            uniq.append(v)
    with open(OUT, "w", encoding="utf-8") as fh:
        json.dump(uniq, fh, indent=1, sort_keys=True, default=str)
    print("%d vectors (%d before de-duplication) -> %s" % (len(uniq), len(vectors), OUT))


if __name__ == "__main__":
    main()
