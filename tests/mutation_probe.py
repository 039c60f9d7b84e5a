"""Differential probe shared by the CPU tier (kernel source on the host simulator) and the GPU tier: the reference's
engine-case inputs, each MUTATED the way an unvalidated in-process caller could hand them to ``engine.Check``
(the C ABI "must re-validate or tolerate violations"), through the dict path and the bytes-in /
bytes-out device road, against ``oracle/check.py`` under several evaluation modes.

An input the device path flags (UNSUPPORTED / left to the host) is counted, never compared; any other difference
from the oracle is a failure.  ``principal.id == ""`` is the mutation the round-3 review found answered differently
(``Index.Query`` drops the principal dimension for an empty id: index/index.go:228-234)."""
import copy
import json

from cerbos_amd import wire
from helpers import load_json, norm_actions
from oracle.check import EvalParams, RuleTableOracle

NOW = 1_700_000_000_000_000_000

MODES = [
    dict(),
    dict(lenient_scope_search=True),
    dict(strict_evaluation=True),
    dict(default_policy_version="20210210"),
    dict(default_scope="acme.hr", lenient_scope_search=True),
]


def _set_attr(inp, side, key, value):
    inp[side].setdefault("attr", {})[key] = value


def mutations(inp):
    """(name, mutated input) pairs; every mutation keeps the message encodable."""
    def m(name):
        c = copy.deepcopy(inp)
        return name, c

    n, c = m("empty_principal_id"); c["principal"]["id"] = ""; yield n, c
    n, c = m("zero_roles"); c["principal"]["roles"] = []; yield n, c
    n, c = m("empty_actions"); c["actions"] = []; yield n, c
    n, c = m("duplicate_actions"); c["actions"] = list(c["actions"]) + list(c["actions"])[::-1]; yield n, c
    n, c = m("many_actions"); c["actions"] = list(c["actions"]) + ["act:%d" % i for i in range(70)]; yield n, c
    n, c = m("duplicate_roles"); c["principal"]["roles"] = list(c["principal"].get("roles") or []) * 2; yield n, c
    n, c = m("many_roles"); c["principal"]["roles"] = ["r%d" % i for i in range(21)] + list(c["principal"].get("roles") or []); yield n, c
    n, c = m("star_role"); c["principal"]["roles"] = ["*"] + list(c["principal"].get("roles") or []); yield n, c
    n, c = m("unknown_kind"); c["resource"]["kind"] = "no_such_kind"; yield n, c
    n, c = m("unknown_version"); c["resource"]["policyVersion"] = "nope"; c["principal"]["policyVersion"] = "nope"; yield n, c
    n, c = m("empty_versions"); c["resource"]["policyVersion"] = ""; c["principal"]["policyVersion"] = ""; yield n, c
    for i, bad in enumerate(("acme..hr", ".acme", "acme.", "acme.hr.uk.london.x", "ACME")):
        n, c = m("odd_scope_%d" % i); c["resource"]["scope"] = bad; c["principal"]["scope"] = bad; yield n, c
    n, c = m("split_scopes"); c["resource"]["scope"] = "acme.hr.uk"; c["principal"]["scope"] = "acme"; yield n, c
    n, c = m("long_action"); c["actions"] = list(c["actions"]) + ["x" * 300]; yield n, c
    n, c = m("non_ascii"); c["actions"] = list(c["actions"]) + ["vïew:ж"]; c["principal"]["roles"] = list(c["principal"].get("roles") or []) + ["rôle"]; yield n, c
    for side in ("principal", "resource"):
        keys = list((inp[side].get("attr") or {}).keys())
        for k in keys[:4]:
            for tag, v in (("null", None), ("string", "zzz"), ("bool", True), ("number", 7), ("nested", {"a": [1, {"b": None}]}),
                           ("list", ["x", 1, False]), ("big", 2 ** 53 + 1), ("neg_big", -(2 ** 63)), ("frac", 0.1),
                           ("unicode", "naïve ✓")):
                n, c = m("%s_attr_%s_%s" % (side, k, tag)); _set_attr(c, side, k, v); yield n, c
            n, c = m("%s_attr_%s_dropped" % (side, k)); del c[side]["attr"][k]; yield n, c
        n, c = m("%s_no_attr" % side); c[side].pop("attr", None); yield n, c
    n, c = m("empty_resource_id"); c["resource"]["id"] = ""; yield n, c
    n, c = m("principal_id_is_other"); c["principal"]["id"] = "daffy_duck" if c["principal"]["id"] != "daffy_duck" else "donald_duck"; yield n, c


def reference_inputs():
    seen, out = set(), []
    for case in load_json("engine_cases.json"):
        for inp in case["inputs"]:
            key = json.dumps(inp, sort_keys=True)
            if key not in seen:
                seen.add(key)
                out.append(inp)
    return out


def _oracle_out(orc, inp, mode, globals_):
    return orc.check(inp, EvalParams(globals_=globals_, now_ns=NOW, **mode))


def run_probe(evaluator, rt, globals_, inputs, roads=("dict", "bytes"), modes=MODES, only=None, chunk=256):
    """Returns {"probes", "flagged", "wrong": [(road, mode, name, have, want)]}."""
    orc = RuleTableOracle(rt)
    probes = flagged = 0
    wrong = []
    pool = []
    for inp in inputs:
        for name, c in mutations(inp):
            if only is None or only(name):
                pool.append((name, c))
    for mode in modes:
        wants = [_oracle_out(orc, c, mode, globals_) for _n, c in pool]
        for road in roads:
            for lo in range(0, len(pool), chunk):
                part = pool[lo:lo + chunk]
                if road == "dict":
                    outs, bad = evaluator.check([c for _n, c in part], now_ns=NOW, allow_unsupported=True, **mode)
                    bad = set(bad)
                else:
                    data, off = wire.pack_messages([wire.encode_check_input(c) for _n, c in part])
                    raw, fl = evaluator.check_pb(data, off, now_ns=NOW, **mode)
                    outs = [wire.decode_check_output(r) for r in raw]
                    bad = {i for i, f in enumerate(fl) if f & 1}
                for i, ((name, c), have) in enumerate(zip(part, outs)):
                    probes += 1
                    if i in bad:
                        flagged += 1
                        continue
                    want = wants[lo + i]
                    if (norm_actions(have) != norm_actions(want)
                            or sorted(have.get("effectiveDerivedRoles") or []) != sorted(want.get("effectiveDerivedRoles") or [])):
                        wrong.append((road, json.dumps(mode, sort_keys=True), name, norm_actions(have), norm_actions(want)))
    return {"probes": probes, "flagged": flagged, "wrong": wrong}
