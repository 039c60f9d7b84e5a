"""Shared test helpers (golden loading, output normalisation)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# now used by the reference's CEL KATs (internal/engine/evaluator_test.go:26)
CEL_EVAL_NOW_NS = 1619103920021000000  # 2021-04-22T10:05:20.021-05:00


def load_json(name):
    with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
        return json.load(f)


def store_rule_table():
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies
    return rule_table_from_policies(policies_from_docs(load_json("store_policies.json")))


def norm_actions(out):
    """{action: (effect, policy, scope)} from a CheckOutput-shaped dict."""
    res = {}
    for a, e in (out.get("actions") or {}).items():
        res[a] = (e.get("effect", "EFFECT_UNSPECIFIED"), e.get("policy", ""), e.get("scope", ""))
    return res
