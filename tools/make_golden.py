#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the reference's own test fixtures.

Run in the build container (needs /root/reference; the GPU box does not have it):

    python tools/make_golden.py

Sources (data fixtures, not code):
  internal/test/testdata/store/**                       policies used by every engine case
  internal/test/testdata/engine/case_*.yaml             TestCheck golden cases (engine_test.go:46-107)
  internal/test/testdata/engine_strict_scope_search/*   (engine_test.go:63)
  internal/test/testdata/engine_lenient_scope_search/*  (engine_test.go:153-210)
  internal/test/testdata/cel_eval/*.yaml                TestSatisfiesCondition KATs (evaluator_test.go:22-48)
  internal/test/testdata/server/checks/check_resources/cr_case_*.yaml   service-level CheckResources cases
                                                        (server_test.go; svc/cerbos_svc.go:274-343), the ones
                                                        that need neither JWT verification nor schema rejection
  internal/engine/testdata/policy_template.yaml.gotmpl  BenchmarkEvaluator policy family (rendered for N=0..1)

The YAML is parsed and re-emitted as compact JSON (of the decision logs only the keys of
auditTrail.effectivePolicies are kept: audit is out of scope, which policies a call touched is
not), so the fixtures travel to the GPU box without the reference tree.
"""
from __future__ import annotations

import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cerbos_amd.policy.loader import load_policy_dir, load_yaml  # noqa: E402

REF = "/root/reference"
TD = os.path.join(REF, "internal/test/testdata")
OUT = os.path.join(ROOT, "tests/golden")


def dump(name, obj):
    path = os.path.join(OUT, name)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(obj, f, sort_keys=True, separators=(",", ":"), ensure_ascii=False)
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


def _camel(k):
    return re.sub(r"_([a-z])", lambda m: m.group(1).upper(), k)


def _norm_msg(d, nested=()):
    """protojson accepts snake_case or camelCase field names; normalise proto-level keys
    (never keys inside attr / jwt maps) to camelCase."""
    out = {}
    for k, v in d.items():
        ck = _camel(k)
        if ck in nested and isinstance(v, dict):
            v = _norm_msg(v)
        out[ck] = v
    return out


def _norm_input(i):
    i = _norm_msg(i, nested=("principal", "resource", "auxData"))
    return i


def _norm_output(o):
    o = _norm_msg(o)
    return o


def engine_cases(subdir, lenient):
    out = []
    for p in sorted(glob.glob(os.path.join(TD, subdir, "*.yaml"))):
        with open(p, encoding="utf-8") as f:
            doc = load_yaml(f.read())
        out.append({
            "name": "%s/%s" % (subdir, os.path.basename(p)[:-5]),
            "description": doc.get("description", ""),
            "lenient": lenient,
            "wantError": bool(doc.get("wantError", False)),
            "inputs": [_norm_input(i) for i in doc.get("inputs") or []],
            "wantOutputs": [_norm_output(o) for o in doc.get("wantOutputs") or []],
            # the call's AuditTrail.EffectivePolicies (engine.go:242-287; check.go:302-304): the policy keys only - the attributes
            # (driver, source file) are the store's
            "wantEffectivePolicies": sorted({k for log in doc.get("wantDecisionLogs") or []
                                            for k in ((log.get("auditTrail") or {}).get("effectivePolicies") or {})}),
            "hasDecisionLogs": bool(doc.get("wantDecisionLogs")),
        })
    return out


def server_check_cases():
    """CheckResourcesRequest cases -> one CheckInput per resource entry (cerbos_svc.go:274-287) with the wanted
    per-action effects and, where the case asks for meta, matched policy / scope and effective derived roles
    (cerbos_svc.go:297-343).  Skipped: cases whose auxData carries JWT tokens (verification is outside the path)
    and cases where the test server's schema enforcement rejects an input (validationErrors)."""
    out = []
    for p in sorted(glob.glob(os.path.join(TD, "server/checks/check_resources", "cr_case_*.yaml"))):
        with open(p, encoding="utf-8") as f:
            raw = f.read()
        doc = load_yaml(raw)
        cr = doc.get("checkResources") or {}
        req, want = cr.get("input") or {}, cr.get("wantResponse") or {}
        if "token" in raw or "validationErrors" in raw or not want.get("results"):
            continue
        inputs, wants = [], []
        for entry, res in zip(req["resources"], want["results"]):
            inp = {"requestId": req.get("requestId", ""), "principal": req["principal"], "resource": entry["resource"],
                   "actions": entry["actions"]}
            inputs.append(_norm_input(inp))
            meta = res.get("meta") or {}
            wants.append({"actions": res["actions"],
                          "meta": {a: {"matchedPolicy": m.get("matchedPolicy", ""), "matchedScope": m.get("matchedScope", "")}
                                   for a, m in (meta.get("actions") or {}).items()},
                          "effectiveDerivedRoles": meta.get("effectiveDerivedRoles"),
                          "hasMeta": bool(meta),
                          # ResultEntry.outputs = CheckOutput.Outputs (cerbos_svc.go:325-327)
                          "outputs": res.get("outputs") or []})
        out.append({"name": "check_resources/%s" % os.path.basename(p)[:-5], "description": doc.get("description", ""),
                    "inputs": inputs, "want": wants})
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    pols = load_policy_dir(os.path.join(TD, "store"))
    dump("store_policies.json", [pols[k] for k in sorted(pols)])

    cases = engine_cases("engine", None)  # run in both scope-search modes
    cases += engine_cases("engine_strict_scope_search", False)
    cases += engine_cases("engine_lenient_scope_search", True)
    dump("engine_cases.json", cases)

    cel = []
    for p in sorted(glob.glob(os.path.join(TD, "cel_eval", "*.yaml"))):
        with open(p, encoding="utf-8") as f:
            doc = load_yaml(f.read())
        cel.append({"name": os.path.basename(p)[:-5], "condition": doc["condition"],
                    "request": doc["request"], "want": bool(doc.get("want", False)),
                    "wantError": bool(doc.get("wantError", False))})
    dump("cel_eval_cases.json", cel)
    dump("server_check_cases.json", server_check_cases())


if __name__ == "__main__":
    main()
