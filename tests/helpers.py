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


def assert_server_case(case, outs, skip=()):
    """A CheckOutput per input against a service-level CheckResources case (tests/golden/server_check_cases.json):
    results[i].actions = the effects; meta.actions[a].matchedPolicy / matchedScope = ActionEffect.policy / scope;
    meta.effectiveDerivedRoles (cerbos_svc.go:297-343).  Returns the number of outputs compared."""
    n = 0
    for i, (have, want) in enumerate(zip(outs, case["want"])):
        if i in skip:
            continue
        assert {a: e["effect"] for a, e in have["actions"].items()} == want["actions"], (case["name"], i)
        for a, m in want["meta"].items():
            assert have["actions"][a]["policy"] == m["matchedPolicy"], (case["name"], i, a)
            if m["matchedScope"] is not None:   # (None: the response has no place for it - PlaygroundEvaluate's EvalResult)
                assert have["actions"][a].get("scope", "") == m["matchedScope"], (case["name"], i, a)
        if want["hasMeta"]:
            assert sorted(have["effectiveDerivedRoles"]) == sorted(want["effectiveDerivedRoles"] or []), (case["name"], i)
        n += 1
    return n


def rfc3339_ns(text):
    """'2022-08-02T15:00:00Z' -> ns since the epoch (test-suite `options.now`)."""
    import datetime
    dt = datetime.datetime.fromisoformat(text.replace("Z", "+00:00"))
    return int(dt.timestamp()) * 1_000_000_000 + dt.microsecond * 1000
