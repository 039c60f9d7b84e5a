"""Writes a synthetic BASELINE workload where the reference can read it: policies as YAML files for a disk store,
CheckInputs as protojson lines.  Input of integration/go/engine_workload_bench_test.go (the Go CPU baseline a
maintainer runs where a Go toolchain exists - SURVEY.md §8(d)).

    python tools/export_workload.py C2 /tmp/c2 [n_requests]
"""
import json
import os
import sys

import yaml

sys.path.insert(0, ".")
from cerbos_amd import workloads  # noqa: E402

WORKLOADS = {"C1": (workloads.c1_policies, workloads.c1_requests), "C2": (workloads.c2_policies, workloads.c2_requests),
             "C3": (workloads.c3_policies, workloads.c3_requests), "C4": (workloads.c4_policies, workloads.c4_requests),
             "C5": (workloads.c5_policies, workloads.c5_requests)}


def export(name, out_dir, n_requests=None):
    pol, reqs = WORKLOADS[name]
    os.makedirs(os.path.join(out_dir, "policies"), exist_ok=True)
    docs = pol()
    for i, doc in enumerate(docs):
        with open(os.path.join(out_dir, "policies", "policy_%05d.yaml" % i), "w") as fh:
            yaml.safe_dump(doc, fh, sort_keys=False)
    cr = reqs(n_requests) if n_requests else reqs()
    n = 0
    with open(os.path.join(out_dir, "inputs.jsonl"), "w") as fh:
        for start in range(0, cr.n, 50_000):
            for inp in cr.to_inputs(start, min(cr.n, start + 50_000)):
                fh.write(json.dumps(inp, separators=(",", ":")) + "\n")
                n += 1
    return len(docs), n


if __name__ == "__main__":
    name, out_dir = sys.argv[1], sys.argv[2]
    n_req = int(sys.argv[3]) if len(sys.argv) > 3 else None
    print("%d policies, %d inputs -> %s" % (export(name, out_dir, n_req) + (out_dir,)))
