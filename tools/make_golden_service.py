#!/usr/bin/env python3
"""tests/golden/service_more_cases.json: the reference's OTHER service-level cases that end in engine.Check - what its server tests
(internal/server/server_test.go:129, LoadTestCases "checks", "playground") hold beyond check_resources/cr_case_* without a token
(tools/make_golden.py mines those):

  server/checks/check_resources/cr_case_*      the cases whose auxData carries a JWT (the token replaced by its claims: verification
                                               is the server's business, auxdata.Extract - cerbos_svc.go:262)
  server/checks/check_resource_set/crs_case_*  CheckResourceSet: one CheckInput per resource instance (cerbos_svc.go:147-166)
  server/checks/check_resource_batch/crb_case_* CheckResourceBatch: one per entry (cerbos_svc.go:213-227)
  server/playground/proxy/pgp_{cr,crs,crb}_case_*  the same three requests against the policies the REQUEST brings - a few files of
                                               the store (playground_svc.go:188-240, an ephemeral engine: no globals, schema
                                               enforcement "warn")
  server/playground/evaluate/pge_case_*        PlaygroundEvaluate: one CheckInput (playground_svc.go:131-186)
  server/authzen/access_evaluation{,_batch}/*  AuthZEN: subject / resource / action -> CheckInput (authzen_svc.go:52-73, 396-425), the
                                               decision = (effect == ALLOW)

Every case in the shape of server_check_cases.json - inputs (CheckInputs), want (per input: effects, matched policy / scope where the
response carries meta, effective derived roles) - plus `globals` and, for the playground's, `policies` (the documents of the files the
request names).  Left out: results the main server's schema enforcement ("reject") decided (validationErrors: schema validation is
outside the path) and the invalid_* cases (request validation).       python tools/make_golden_service.py
"""
from __future__ import annotations

import base64
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cerbos_amd.policy.loader import load_yaml  # noqa: E402
from tools.make_golden import _norm_input  # noqa: E402

TD = "/root/reference/internal/test/testdata"
OUT = os.path.join(ROOT, "tests/golden/service_more_cases.json")
MAIN_GLOBALS = {"environment": "test"}   # the test server's engine (server_test.go, as the engine cases)


def claims(tok):
    payload = "".join(tok.split()).split(".")[1]
    return json.loads(base64.urlsafe_b64decode(payload + "=" * (-len(payload) % 4)))


def aux_of(req):
    """Request AuxData (tokens) -> engine AuxData (claims), or None."""
    aux = req.get("auxData") or {}
    out = {}
    # (a keySetId names the key set to verify with: nothing of it reaches the engine)
    if (aux.get("jwt") or {}).get("token"):
        out["jwt"] = claims(aux["jwt"]["token"])
    if aux.get("jwts"):   # named tokens: enginev1.AuxData.jwts, name -> {claims}
        out["jwts"] = {k: {"claims": claims(v["token"])} for k, v in aux["jwts"].items()}
    return out or None


def want_of(actions, meta_actions=None, edr=None, has_meta=False, outputs=None):
    return {"actions": actions,
            "meta": {a: {"matchedPolicy": m.get("matchedPolicy", ""), "matchedScope": m.get("matchedScope", "")}
                     for a, m in (meta_actions or {}).items()},
            "effectiveDerivedRoles": edr, "hasMeta": has_meta, "outputs": outputs or []}


def from_check_resources(req, want):
    inputs, wants = [], []
    aux = aux_of(req)
    for entry, res in zip(req["resources"], want.get("results") or []):   # (no results: the response is a failure - a policy that does not compile)
        if res.get("validationErrors"):
            continue
        inp = {"requestId": req.get("requestId", ""), "principal": req["principal"], "resource": entry["resource"], "actions": entry["actions"]}
        if aux:
            inp["auxData"] = aux
        meta = res.get("meta") or {}
        inputs.append(_norm_input(inp))
        wants.append(want_of(res["actions"], meta.get("actions"), meta.get("effectiveDerivedRoles"), bool(meta), res.get("outputs")))
    return inputs, wants


def from_check_resource_set(req, want):
    inputs, wants = [], []
    aux = aux_of(req)
    rs = req["resource"]
    meta_all = (want.get("meta") or {}).get("resourceInstances") or {}
    for key, inst in rs["instances"].items():
        res = (want.get("resourceInstances") or {}).get(key)
        if res is None or res.get("validationErrors"):
            continue
        resource = {"kind": rs["kind"], "id": key, "attr": (inst or {}).get("attr") or {}}
        for k in ("policyVersion", "scope"):
            if rs.get(k):
                resource[k] = rs[k]
        inp = {"requestId": req.get("requestId", ""), "principal": req["principal"], "resource": resource, "actions": req["actions"]}
        if aux:
            inp["auxData"] = aux
        meta = meta_all.get(key) or {}
        inputs.append(_norm_input(inp))
        wants.append(want_of(res["actions"], meta.get("actions"), meta.get("effectiveDerivedRoles"), bool(want.get("meta"))))
    return inputs, wants


def from_check_resource_batch(req, want):
    inputs, wants = [], []
    aux = aux_of(req)
    for entry, res in zip(req["resources"], want.get("results") or []):
        if res.get("validationErrors"):
            continue
        inp = {"requestId": req.get("requestId", ""), "principal": req["principal"], "resource": entry["resource"], "actions": entry["actions"]}
        if aux:
            inp["auxData"] = aux
        inputs.append(_norm_input(inp))
        wants.append(want_of(res["actions"]))
    return inputs, wants


def _cp(props, k):
    v = (props or {}).get("cerbos." + k)
    return v if isinstance(v, str) else ""


def authzen_input(subject, resource, action, context):
    """authzen_svc.go:52-73, 396-425: subject / resource / action -> CheckInput.  The properties ARE the attributes (the cerbos.* ones
    among them); roles = the strings of cerbos.roles, else the subject's type."""
    sp, rp = subject.get("properties") or {}, resource.get("properties") or {}
    roles = [r for r in (sp.get("cerbos.roles") or []) if isinstance(r, str) and r] if isinstance(sp.get("cerbos.roles"), list) else []
    principal = {"id": subject.get("id", ""), "roles": roles or [subject.get("type", "")], "attr": sp}
    res = {"kind": resource.get("type", ""), "id": resource.get("id", ""), "attr": rp}
    for d, props in ((principal, sp), (res, rp)):
        for k in ("policyVersion", "scope"):
            if _cp(props, k):
                d[k] = _cp(props, k)
    inp = {"requestId": _cp(context, "requestId"), "principal": principal, "resource": res, "actions": [action.get("name", "")]}
    aux = aux_of({"auxData": (context or {}).get("cerbos.auxData")})
    if aux:
        inp["auxData"] = aux
    return _norm_input(inp)


FILE_RE = re.compile(r"fileString\s+`([^`]+)`")


def playground_policies(files):
    docs = []
    for f in files:
        m = FILE_RE.search(f.get("contents", ""))
        if not m:
            raise SystemExit("playground file without a fileString template: %r" % f)
        rel = m.group(1)
        if "/_schemas/" in "/" + rel or rel.endswith(".json"):
            continue
        with open(os.path.join(TD, rel), encoding="utf-8") as fh:
            docs.append(load_yaml(fh.read()))
    return docs


def main():
    cases = []

    def add(name, doc, inputs, wants, globals_, policies=None):
        if not inputs:
            return
        c = {"name": name, "description": doc.get("description", ""), "inputs": inputs, "want": wants, "globals": globals_}
        if policies is not None:
            c["policies"] = policies
        cases.append(c)

    def load(p):
        with open(p, encoding="utf-8") as f:
            raw = f.read()
        return raw, load_yaml(raw)

    for p in sorted(glob.glob(os.path.join(TD, "server/checks/check_resources", "cr_case_*.yaml"))):
        raw, doc = load(p)
        cr = doc.get("checkResources") or {}
        req, want = cr.get("input") or {}, cr.get("wantResponse") or {}
        if "token" not in raw or not want.get("results"):   # (the others: server_check_cases.json)
            continue
        add("check_resources/%s" % os.path.basename(p)[:-5], doc, *from_check_resources(req, want), MAIN_GLOBALS)
    for p in sorted(glob.glob(os.path.join(TD, "server/checks/check_resource_set", "crs_case_*.yaml"))):
        raw, doc = load(p)
        c = doc.get("checkResourceSet") or {}
        add("check_resource_set/%s" % os.path.basename(p)[:-5], doc, *from_check_resource_set(c.get("input") or {}, c.get("wantResponse") or {}), MAIN_GLOBALS)
    for p in sorted(glob.glob(os.path.join(TD, "server/checks/check_resource_batch", "crb_case_*.yaml"))):
        raw, doc = load(p)
        c = doc.get("checkResourceBatch") or {}
        add("check_resource_batch/%s" % os.path.basename(p)[:-5], doc, *from_check_resource_batch(c.get("input") or {}, c.get("wantResponse") or {}), MAIN_GLOBALS)
    for p in sorted(glob.glob(os.path.join(TD, "server/playground/proxy", "pgp_c*_case_*.yaml"))):
        if not re.match(r"pgp_(cr|crs|crb)_case_\d+\.yaml$", os.path.basename(p)):   # (not the invalid_* ones)
            continue
        raw, doc = load(p)
        c = doc.get("playgroundProxy") or {}
        req, want = c.get("input") or {}, c.get("wantResponse") or {}
        pols = playground_policies(req.get("files") or [])
        name = "playground_proxy/%s" % os.path.basename(p)[:-5]
        if "checkResources" in req:
            add(name, doc, *from_check_resources(req["checkResources"], want.get("checkResources") or {}), {}, pols)
        elif "checkResourceSet" in req:
            add(name, doc, *from_check_resource_set(req["checkResourceSet"], want.get("checkResourceSet") or {}), {}, pols)
        elif "checkResourceBatch" in req:
            add(name, doc, *from_check_resource_batch(req["checkResourceBatch"], want.get("checkResourceBatch") or {}), {}, pols)
    for p in sorted(glob.glob(os.path.join(TD, "server/playground/evaluate", "pge_case_*.yaml"))):
        raw, doc = load(p)
        c = doc.get("playgroundEvaluate") or {}
        req, want = c.get("input") or {}, (c.get("wantResponse") or {}).get("success")
        if not want:
            continue
        inp = {"requestId": req.get("playgroundId", ""), "principal": req["principal"], "resource": req["resource"], "actions": req["actions"]}
        aux = aux_of(req)
        if aux:
            inp["auxData"] = aux
        acts = {r["action"]: r["effect"] for r in want["results"]}
        meta = {r["action"]: {"matchedPolicy": r.get("policy", "")} for r in want["results"]}
        w = want_of(acts, None, want.get("effectiveDerivedRoles"), True, want.get("outputs"))
        w["meta"] = {a: {"matchedPolicy": m["matchedPolicy"], "matchedScope": None} for a, m in meta.items()}   # (EvalResult carries no scope)
        add("playground_evaluate/%s" % os.path.basename(p)[:-5], doc, [_norm_input(inp)], [w], {}, playground_policies(req.get("files") or []))
    # AuthZEN (authzen_svc.go): one evaluation = one CheckInput with one action; decision = (effect == ALLOW).  A batch merges every
    # evaluation with the request's defaults (a member given replaces the default whole, authzen_svc.go:342-347), decides all of them
    # and cuts the answers behind the first deny / permit its semantic names (:281-288).  (The service groups the evaluations of one
    # subject and resource into one CheckInput: the actions of an input are decided independently - same decisions.)
    for p in sorted(glob.glob(os.path.join(TD, "server/authzen/access_evaluation", "ae_case_*.yaml"))):
        raw, doc = load(p)
        c = doc.get("accessEvaluation") or {}
        req, want = c.get("input") or {}, c.get("wantResponse") or {}
        inp = authzen_input(req["subject"], req["resource"], req["action"], req.get("context"))
        resp = ((want.get("context") or {}).get("cerbos.response") or {}).get("results") or [{}]
        meta = resp[0].get("meta") or {}
        w = want_of({inp["actions"][0]: "EFFECT_ALLOW" if want.get("decision") else "EFFECT_DENY"}, None, meta.get("effectiveDerivedRoles"), bool(meta))
        w["meta"] = {a: {"matchedPolicy": m.get("matchedPolicy", ""), "matchedScope": m.get("matchedScope", "")} for a, m in (meta.get("actions") or {}).items()}
        add("authzen/%s" % os.path.basename(p)[:-5], doc, [inp], [w], MAIN_GLOBALS)
    for p in sorted(glob.glob(os.path.join(TD, "server/authzen/access_evaluation_batch", "aeb_case_*.yaml"))):
        raw, doc = load(p)
        c = doc.get("accessEvaluationBatch") or {}
        req, want = c.get("input") or {}, c.get("wantResponse") or {}
        inputs, wants = [], []
        for ev, res in zip(req.get("evaluations") or [], want.get("evaluations") or []):   # (the answers the semantic left)
            ctx = ev.get("context") or req.get("context")
            inp = authzen_input(ev.get("subject") or req.get("subject"), ev.get("resource") or req.get("resource"), ev.get("action") or req.get("action"), ctx)
            inputs.append(inp)
            wants.append(want_of({inp["actions"][0]: "EFFECT_ALLOW" if res.get("decision") else "EFFECT_DENY"}))
        add("authzen_batch/%s" % os.path.basename(p)[:-5], doc, inputs, wants, MAIN_GLOBALS)
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(cases, f, sort_keys=True, separators=(",", ":"), ensure_ascii=False)
        f.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(cases), "cases,", sum(len(c["inputs"]) for c in cases), "inputs")
    for c in cases:
        print("  %-48s %2d inputs%s" % (c["name"], len(c["inputs"]), ("  own policies: %d" % len(c["policies"])) if "policies" in c else ""))


if __name__ == "__main__":
    main()
