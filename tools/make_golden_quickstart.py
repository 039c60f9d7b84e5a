"""Mines the reference's quickstart (docs/modules/ROOT/pages/quickstart.adoc + examples/quickstart/curl.txt): one
CheckResourcesRequest, three stages of the policy directory (empty; a derived-roles file and a resource policy; the policy with one
more rule) and the CheckResourcesResponse the documentation publishes after each -> tests/golden/quickstart.json.

    python tools/make_golden_quickstart.py        (needs /root/reference; the GPU box never runs this)"""
import json
import os
import re

import yaml

DOCS = "/root/reference/docs/modules/ROOT"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "quickstart.json")


def main():
    curl = open(os.path.join(DOCS, "examples/quickstart/curl.txt"), encoding="utf-8").read()
    request = json.loads(curl[curl.index("{"):curl.rindex("}") + 1])
    text = open(os.path.join(DOCS, "pages/quickstart.adoc"), encoding="utf-8").read()
    # walk the page in order: `cat > cerbos-quickstart/policies/<file> <<EOF ... EOF` writes a policy file, a `.Response` block
    # closes a stage with the directory as it stands
    events = []
    for m in re.finditer(r"cat > cerbos-quickstart/policies/(\S+) <<EOF\n(.*?)\nEOF", text, re.S):
        events.append((m.start(), "file", m.group(1), m.group(2)))
    for m in re.finditer(r"\.Response\n\[source,json\]\n----\n(.*?)\n----", text, re.S):
        events.append((m.start(), "response", None, m.group(1)))
    events.sort()
    files, stages = {}, []
    for _at, kind, name, body in events:
        if kind == "file":
            files[name] = yaml.safe_load(body)
        else:
            stages.append({"policies": [files[k] for k in sorted(files)], "files": sorted(files), "response": json.loads(body)})
    doc = {"source": "docs/modules/ROOT/pages/quickstart.adoc, docs/modules/ROOT/examples/quickstart/curl.txt", "request": request, "stages": stages}
    json.dump(doc, open(OUT, "w"), indent=1)
    print(len(stages), "stages;", [s["files"] for s in stages])


if __name__ == "__main__":
    main()
