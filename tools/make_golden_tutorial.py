"""Mines the reference's tutorial (docs/modules/ROOT/examples/tutorial/*/cerbos): every stage is a policy directory with policy-test
suites (tests/*_test.yaml: named principals and resources, per test the principals x resources x actions asked for and the effects
expected) that the reference's own `cerbos compile` runs (test.sh).  -> tests/golden/tutorial_suites.json: per stage the policy
documents and one vector per expected (principal, resource) entry: the CheckInput and the effects.

    python tools/make_golden_tutorial.py        (needs /root/reference; the GPU box never runs this)"""
import glob
import json
import os

import yaml

ROOT = "/root/reference/docs/modules/ROOT/examples/tutorial"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tutorial_suites.json")


def main():
    stages = []
    for d in sorted(glob.glob(os.path.join(ROOT, "*", "cerbos"))):
        policies = []
        for f in sorted(glob.glob(os.path.join(d, "policies", "*.yaml"))):
            policies.extend(doc for doc in yaml.safe_load_all(open(f, encoding="utf-8")) if doc)
        vectors = []
        for f in sorted(glob.glob(os.path.join(d, "tests", "*_test.yaml"))):
            suite = yaml.safe_load(open(f, encoding="utf-8"))
            for t in suite.get("tests") or []:
                if t.get("skip"):
                    continue
                actions = list(t["input"]["actions"])
                for e in t.get("expected") or []:
                    assert e["principal"] in t["input"]["principals"] and e["resource"] in t["input"]["resources"]
                    p, r = suite["principals"][e["principal"]], suite["resources"][e["resource"]]
                    want = {a: e["actions"].get(a, "EFFECT_DENY") for a in actions}   # an action left out is expected to be denied
                    vectors.append({"suite": os.path.basename(f), "test": t["name"], "principal": e["principal"], "resource": e["resource"],
                                    "input": {"requestId": "%s/%s" % (e["principal"], e["resource"]), "principal": p, "resource": r, "actions": actions},
                                    "want": want})
        if vectors:
            stages.append({"stage": os.path.basename(os.path.dirname(d)), "policies": policies, "vectors": vectors})
    json.dump({"source": "docs/modules/ROOT/examples/tutorial/*/cerbos/{policies,tests}", "stages": stages}, open(OUT, "w"), indent=1)
    print([(s["stage"], len(s["policies"]), len(s["vectors"])) for s in stages])


if __name__ == "__main__":
    main()
