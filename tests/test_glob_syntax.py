"""Glob syntax beyond `*`: `? [] [!] {} \\` (gobwas/glob v0.2.3 with separator ':', the dialect the reference
compiles in internal/util/globs_common.go:31) through the device automaton.

The reference has no direct glob unit test (parity-unpinned syntax), so the anchor is
oracle/globmatch.py's restatement of the documented pattern language.  Index dimensions treat a key as a
pattern only when it contains `*` (glob_dimension.go:32): every rule pattern below carries one; role-policy
allow-lists go through MatchesGlob unconditionally (index.go:447-452), so those patterns need not.

CPU tier: the lowering's host simulation of the automaton.  GPU tier: cbh_resolve_globs_kernel resolves the
batch-local strings on the device and the decision kernel consumes the bits - compared per action with the
oracle's decision.
"""

import numpy as np
import pytest

from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle
from oracle.globmatch import fix_glob, glob_match

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000

ACTION_GLOBS = ["v?ew:*", "[ab]*", "[!ab]dit*", "{read,write}:*", "x\\*y*", "a:**:z*", "[a-c]?:*:{x,yy}*", "*:?", "é?*",
                "{a,b{c,d}}:*"]
ROLE_GLOBS = ["man?ger*", "[tu]ser*", "{dev,ops}_*"]
ALLOW_GLOBS = ["v?ew", "[ab]c", "{read,write}:doc", "lit\\*eral", "x:*"]   # role-policy allow-lists: no `*` needed


def _strings(rng, n):
    """Strings built from the pieces the patterns mention, so that near misses are frequent."""
    pieces = ["v", "i", "e", "w", "view", "vxew", ":", "a", "b", "c", "d", "dit", "edit", "read", "write", "x", "*", "y",
              "z", "é", "ü", "yy", "doc", "", "q", "man", "ager", "manager", "manxger", "tser", "user", "dev_", "ops_",
              "lit", "eral", "ac", "bc"]
    out = set()
    while len(out) < n:
        k = int(rng.integers(1, 6))
        out.add("".join(str(rng.choice(pieces)) for _ in range(k)))
    out.discard("")
    return sorted(out)


def _docs():
    rules = [{"actions": [g], "effect": "EFFECT_ALLOW", "roles": ["*"], "name": "g%d" % i} for i, g in enumerate(ACTION_GLOBS)]
    rules += [{"actions": ["role:*"], "effect": "EFFECT_ALLOW", "roles": [g], "name": "r%d" % i} for i, g in enumerate(ROLE_GLOBS)]
    return [
        {"apiVersion": API, "resourcePolicy": {"resource": "doc", "version": "default", "rules": rules}},
        {"apiVersion": API, "resourcePolicy": {"resource": "vault", "version": "default", "rules": [
            {"actions": ["*"], "effect": "EFFECT_ALLOW", "roles": ["*"]}]}},
        {"apiVersion": API, "rolePolicy": {"role": "clerk", "rules": [{"resource": "vault", "allowActions": ALLOW_GLOBS}]}},
    ]


def _inputs(rng):
    acts = _strings(rng, 700)
    roles = _strings(rng, 80)
    ins = []
    for i in range(0, len(acts), 7):
        ins.append({"requestId": "a%d" % i, "principal": {"id": "p", "roles": ["someone"]},
                    "resource": {"kind": "doc", "id": "d"}, "actions": acts[i:i + 7]})
    for i, r in enumerate(roles):
        ins.append({"requestId": "r%d" % i, "principal": {"id": "p", "roles": [r]},
                    "resource": {"kind": "doc", "id": "d"}, "actions": ["role:x", "other"]})
    for i in range(0, len(acts), 9):
        ins.append({"requestId": "c%d" % i, "principal": {"id": "p", "roles": ["clerk"]},
                    "resource": {"kind": "vault", "id": "d"}, "actions": acts[i:i + 9] + ["view", "vxew", "ac", "lit*eral", "x:1"]})
    return ins


def _tables():
    rt = rule_table_from_policies(policies_from_docs(_docs()))
    return rt, lower_rule_table(rt)


def test_lowered_automaton_matches_the_oracle_glob():
    _, lt = _tables()
    rng = np.random.default_rng(7)
    strings = _strings(rng, 3000)
    n = 0
    for dim, pats in ((0, ACTION_GLOBS + ALLOW_GLOBS), (1, ROLE_GLOBS)):
        nfa = lt.nfas[dim]
        index = {fix_glob(p): i for i, p in enumerate(nfa.patterns)}
        assert all(p in index for p in pats), (dim, nfa.patterns)
        for s in strings:
            bits = nfa.match_bits(s.encode("utf-8"))
            for p in pats:
                assert bool((bits >> index[p]) & 1) == glob_match(p, s), (p, s)
                n += 1
    assert n > 30_000


def _decisions_agree(ev, rt):
    orc = RuleTableOracle(rt)
    inputs = _inputs(np.random.default_rng(11))
    outs = ev.check(inputs, now_ns=NOW)
    allow = 0
    for inp, have in zip(inputs, outs):
        want = orc.check(inp, EvalParams(now_ns=NOW))
        assert norm_actions(have) == norm_actions(want), inp
        allow += sum(e["effect"] == "EFFECT_ALLOW" for e in want["actions"].values())
    assert allow > 100   # the patterns do fire
    return len(inputs)


def test_kernel_source_decides_exotic_globs_like_the_oracle():
    from test_hostsim_golden import HostSimEvaluator
    rt, lt = _tables()
    assert _decisions_agree(HostSimEvaluator(lt, Conf()), rt) > 100


@pytest.mark.gpu
def test_gpu_resolves_exotic_globs_like_the_oracle():
    rt, lt = _tables()
    ev = HipEvaluator(lt, Conf())
    try:
        assert _decisions_agree(ev, rt) > 100
    finally:
        ev.close()
