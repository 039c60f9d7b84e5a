"""tools/export_workload.py writes the synthetic workloads in the reference's own formats (disk-store YAML,
protojson lines) for the Go baseline of integration/go/: what it writes must reload into the same rule table
and the same inputs."""
import json
import os
import sys

import pytest

from cerbos_amd.policy.loader import load_policy_dir, policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import export_workload  # noqa: E402


@pytest.mark.parametrize("name", ["C2", "C3", "C5"])
def test_export_round_trip(tmp_path, name):
    n_pol, n_in = export_workload.export(name, str(tmp_path), 300)
    pol, reqs = export_workload.WORKLOADS[name]
    assert n_in == 300 and n_pol == len(pol())
    a = rule_table_from_policies(load_policy_dir(str(tmp_path / "policies")))
    b = rule_table_from_policies(policies_from_docs(pol()))
    assert json.dumps(a["rules"], sort_keys=True, default=str) == json.dumps(b["rules"], sort_keys=True, default=str)
    with open(tmp_path / "inputs.jsonl") as fh:
        got = [json.loads(line) for line in fh]
    assert got == reqs(300).to_inputs()
