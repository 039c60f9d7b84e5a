"""The reference's tutorial policy directories and their policy-test suites (docs/modules/ROOT/examples/tutorial/*/cerbos, run by
the reference's own `cerbos compile`; mined by tools/make_golden_tutorial.py into tests/golden/tutorial_suites.json): five stages
- resource policies, conditions, derived roles, a principal policy, attribute schemas (enforcement is off in the engine path this
repository covers: the schemas reference is carried, not enforced) - 97 expected (principal, resource) entries.  Every entry by
the policy-level oracle and by the device path (lowering + the kernel source on the host simulator)."""
import pytest

from cerbos_amd.engine import Conf
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json
from oracle.check import EvalParams, RuleTableOracle
from test_hostsim_golden import HostSimEvaluator

STAGES = load_json("tutorial_suites.json")["stages"]
NOW = 1_700_000_000_000_000_000


@pytest.mark.parametrize("stage", STAGES, ids=lambda s: s["stage"])
def test_tutorial_stage(stage):
    rt = rule_table_from_policies(policies_from_docs(stage["policies"]))
    orc = RuleTableOracle(rt)
    inputs = [v["input"] for v in stage["vectors"]]
    for v in stage["vectors"]:
        have = orc.check(v["input"], EvalParams(now_ns=NOW))
        assert {a: e["effect"] for a, e in have["actions"].items()} == v["want"], (stage["stage"], v["suite"], v["test"], v["principal"], v["resource"])
    outs, bad = HostSimEvaluator(lower_rule_table(rt), Conf()).check(inputs, now_ns=NOW, allow_unsupported=True)
    assert not bad, (stage["stage"], bad)
    for v, have in zip(stage["vectors"], outs):
        assert {a: e["effect"] for a, e in have["actions"].items()} == v["want"], (stage["stage"], v["suite"], v["test"], v["principal"], v["resource"])
