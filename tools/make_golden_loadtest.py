#!/usr/bin/env python3
"""Export the reference's load-test policy / request template sets (BASELINE.json configs[0]:
`hack/loadtest`) as a fixture that travels to the GPU box.

    python tools/make_golden_loadtest.py        # needs /root/reference (build container only)

Source (data templates, not code): hack/loadtest/templates/{classic,multitenant}/{policies,requests,files}.
hack/loadtest/generate.go renders every template `Count` times with
    .NameMod "x"  -> fmt.Sprintf("%s_%05d", x, N)       (generate.go:49-51)
    .RequestID    -> fmt.Sprintf("REQ_%05d", N)          (generate.go:90-93)
and copies `files/` verbatim.  The fixture keeps the template text with those two constructs replaced by
the placeholders  x_@N@  /  REQ_@N@ ; cerbos_amd.workloads.loadtest_set() does the per-N substitution.
Schemas (`schemas/`) are not exported: schema validation is outside the path.
"""
from __future__ import annotations

import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/hack/loadtest/templates"
OUT = os.path.join(ROOT, "tests", "golden", "loadtest_templates.json")

NAMEMOD = re.compile(r"\{\{\s*\.NameMod\s+[\"`]([A-Za-z0-9_]+)[\"`]\s*\}\}")
REQID = re.compile(r"\{\{\s*\.RequestID\s*\}\}")


def placeholders(text):
    text = NAMEMOD.sub(lambda m: m.group(1) + "_@N@", text)
    text = REQID.sub("REQ_@N@", text)
    if "{{" in text:
        raise SystemExit("unhandled template construct in: " + text[:200])
    return text


def read(path):
    with open(path, encoding="utf-8") as f:
        return f.read()


def main():
    out = {}
    for name in ("classic", "multitenant"):
        base = os.path.join(SRC, name)
        out[name] = {
            "policies": [{"file": os.path.basename(p), "text": placeholders(read(p))}
                         for p in sorted(glob.glob(os.path.join(base, "policies", "*.tpl")))],
            "static_policies": [{"file": os.path.basename(p), "text": read(p)}
                                for p in sorted(glob.glob(os.path.join(base, "files", "policies", "*.yaml")))],
            "requests": [{"file": os.path.basename(p), "text": placeholders(read(p))}
                         for p in sorted(glob.glob(os.path.join(base, "requests", "*.tpl")))],
        }
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(out, f, sort_keys=True, separators=(",", ":"))
        f.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
