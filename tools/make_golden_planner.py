#!/usr/bin/env python3
"""Regenerate tests/golden/planner_cases.json from the reference's query-planner fixtures (data, not code):

  internal/test/testdata/query_planner/policies/**     the policies TestQueryPlan loads (engine_test.go:420-422)
  internal/test/testdata/query_planner_filter/*.yaml    TestNormaliseFilter cases (planner_test.go:443-456)
  internal/ruletable/planner/testdata/ast_build_expr.yaml   Test_buildExpr cases (ast_test.go:49-70)
  internal/test/testdata/query_planner/suite/{common,strict_scope_search,lenient_scope_search}/*.yaml
                                                        QueryPlannerTestSuite files: a principal and tests of (action(s), resource,
                                                        wanted filter) - engine_test.go:427-495

Run in the build container (needs /root/reference):   python tools/make_golden_planner.py
YAML anchors / merge keys are resolved by the loader; the wanted filters are kept as written (operand order is free: the
reference's own comparison sorts the operands of every expression, engine_test.go:486-488)."""
from __future__ import annotations

import glob
import json
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cerbos_amd.policy.loader import load_policy_dir  # noqa: E402

TD = "/root/reference/internal/test/testdata/query_planner"
OUT = os.path.join(ROOT, "tests/golden/planner_cases.json")


def main():
    pols = load_policy_dir(os.path.join(TD, "policies"))
    suites = []
    for mode, sub in ((None, "common"), (False, "strict_scope_search"), (True, "lenient_scope_search")):
        for p in sorted(glob.glob(os.path.join(TD, "suite", sub, "*.yaml"))):
            with open(p, encoding="utf-8") as f:
                doc = yaml.safe_load(f.read())
            tests = []
            for t in doc.get("tests") or []:
                tests.append({"actions": t["actions"] if t.get("actions") is not None else [t.get("action", "")],
                              "resource": t.get("resource") or {}, "want": t.get("want") or {}, "wantErr": bool(t.get("wantErr", False))})
            suites.append({"name": "%s/%s" % (sub, os.path.basename(p)[:-5]), "lenient": mode, "description": doc.get("description", ""),
                           "principal": doc.get("principal") or {}, "tests": tests})
    # TestNormaliseFilter (planner_test.go:443-456): filters in, normalised filters and their debug strings out
    filters = []
    for p in sorted(glob.glob("/root/reference/internal/test/testdata/query_planner_filter/*.yaml")):
        with open(p, encoding="utf-8") as f:
            doc = yaml.safe_load(f.read())
        filters.append({"name": os.path.basename(p)[:-5], "description": doc.get("description", ""), "input": doc.get("input") or {},
                        "wantFilter": doc.get("wantFilter") or {}, "wantString": doc.get("wantString", "")})
    # Test_buildExpr (ast_test.go:49-70): CEL text -> the filter operand buildExpr makes of it
    with open("/root/reference/internal/ruletable/planner/testdata/ast_build_expr.yaml", encoding="utf-8") as f:
        build_expr = yaml.safe_load(f.read())
    # the service-level cases (internal/test/testdata/server/plan_resources; svc/cerbos_svc.go:53-118) over the engine store of
    # tests/golden/store_policies.json: PlanResourcesRequest in, PlanResourcesResponse out.  The request's JWT is replaced by its
    # claims (the token's payload, decoded here: verification is the server's business, as on the check path)
    import base64
    server = []
    for p in sorted(glob.glob("/root/reference/internal/test/testdata/server/plan_resources/*.yaml")):
        with open(p, encoding="utf-8") as f:
            doc = yaml.safe_load(f.read())
        pr = doc.get("planResources") or {}
        inp = dict(pr.get("input") or {})
        def claims(tok):
            payload = tok.split(".")[1]
            return json.loads(base64.urlsafe_b64decode(payload + "=" * (-len(payload) % 4)))
        aux = inp.get("auxData") or {}
        if (aux.get("jwt") or {}).get("token") or aux.get("jwts"):
            inp["auxData"] = {}
            if (aux.get("jwt") or {}).get("token"):
                inp["auxData"]["jwt"] = claims(aux["jwt"]["token"])
            if aux.get("jwts"):
                inp["auxData"]["jwts"] = {k: {"claims": claims(v["token"])} for k, v in aux["jwts"].items()}
        server.append({"name": os.path.basename(p)[:-5], "description": doc.get("description", ""), "input": inp,
                       "wantResponse": pr.get("wantResponse") or {}, "wantStatus": doc.get("wantStatus") or {}})
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump({"policies": [pols[k] for k in sorted(pols)], "suites": suites, "filters": filters, "buildExpr": build_expr, "serverPlans": server}, f, sort_keys=True, separators=(",", ":"), ensure_ascii=False)
        f.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(suites), "suites,", sum(len(s["tests"]) for s in suites), "tests")


if __name__ == "__main__":
    main()
