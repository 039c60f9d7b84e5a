"""The reference's naming rules against its own Go-coded tables, mined by tools/make_golden_namer.py:
internal/namer/namer_test.go - TestFQN, TestFQNTree, TestFQNSpecialChars (with the xxhash64 module ids that key a
runtimev1.RuleTable's maps: cerbos_amd/ruletable/proto.py) - and internal/conditions/identifiers_test.go."""
from cerbos_amd import namer
from cerbos_amd.policy import compile as pc
from cerbos_amd.policy.loader import policy_fqn
from cerbos_amd.ruletable.proto import module_id
from helpers import load_json

V = load_json("namer_vectors.json")


def test_fqn():
    for c in V["fqn"]:
        assert policy_fqn(c["policy"]) == c["want"], c["name"]


def test_fqn_tree():
    for c in V["fqn_tree"]:
        assert [policy_fqn(c["policy"])] + pc._ancestor_fqns(c["policy"]) == c["want"], c["name"]


def test_special_characters_and_module_ids():
    for c in V["special_chars"]:
        fn = namer.resource_policy_fqn if c["kind"] == "resource" else namer.principal_policy_fqn
        fqn = fn(c["policyName"], c["version"], c["scope"])
        assert fqn == c["wantFQN"]
        assert str(module_id(fqn)) == c["wantModuleID"], fqn


def test_identifiers():
    for name in V["identifiers"]["valid"]:
        assert pc._identifier_error(name) is None, name
    for name in V["identifiers"]["invalid"]:
        assert pc._identifier_error(name), name
