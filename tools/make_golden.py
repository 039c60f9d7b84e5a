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
  internal/engine/testdata/policy_template.yaml.gotmpl  BenchmarkEvaluator policy family (rendered for N=0..1)

The YAML is parsed and re-emitted as compact JSON (decision logs dropped: audit is out of
scope), so the fixtures travel to the GPU box without the reference tree.
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
        })
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


if __name__ == "__main__":
    main()
