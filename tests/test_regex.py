"""`matches` on the device path: RE2 patterns compiled to byte-level DFAs at lowering time (cerbos_amd/lower/regex.py),
one table lookup per byte on the device (cbh_vm.h regex_match).

1. The automaton against the oracle's RE2 reading (oracle/celeval.py _re: Python `re` behind an RE2 -> Python
   translation) over a corpus of patterns x strings, and over randomly generated patterns and strings.
2. The same through the whole device path - policy conditions `R.attr.s.matches(<pattern>)` - against oracle/check.py:
   CPU tier on the host simulator, GPU tier on the kernel.
Patterns outside the subset (flags, \\b, Unicode classes ...) must be flagged UNSUPPORTED, never answered."""
import numpy as np
import pytest

from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower import regex
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle import celeval
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000
PATTERNS = [r"^[mM].*g$", r"^comm", r"a|b", r"(ab)+c$", r"^$", r"x{2,3}y", r"\d+\.\d*", r"[^a-c]z", r"h.llo", r"^(foo|bar)baz?$",
            r"a{0}b", r"[a-z]+@[a-z]+\.com$", "é+x", ".é.", r"^\s*\S+\s*$", r"(a|ab)(c|bcd)(d*)", r"a{2,}$", r"^.{3}$", r"[\d\-x]+$",
            r"a\$b", r"{x}", r"a{,3}", r"^(?:[a-z0-9_]+\.)*[a-z]+$", r"\Aab\z", r"(?P<n>ab)*c", r"[]a]+b", r"[^]a]b", r"\w+\W\w+", r"\D\d",
            r"^/api/v[12]/(users|groups)/[0-9a-f]{8}$", r"(a*)*b", r"a.*b.*c", r"[a-c]{2}[x-z]?$", r"\x41\x2e", r"colou?r", r"^-?\d+(\.\d+)?$"]
STRINGS = ["marketing", "Marketing", "mg", "communications", "a", "b", "abab", "ababc", "", "xy", "xxy", "xxxxy", "3.14", "3.", "dz", "az",
           "hello", "hallo", "h\nllo", "foobaz", "barba", "foo", "ab", "bob@example.com", "éééx", "aéb", "  word ", "two words", "abcd", "aa",
           "aaa", "aéé", "héé", "1-2x", "a$b", "{x}", "a{,3}", "a\n", "acme.hr.uk", "acme..hr", "abc", "]ab", "xb", "ab cd", "a1", "x7",
           "/api/v1/users/0123abcd", "/api/v3/users/0123abcd", "/api/v2/groups/0123abcg", "aab", "axxbxxc", "acb", "abz", "A.", "color",
           "colour", "-12.50", "12.", "日本語", "a日b", "café"]
OUTSIDE = [r"(?i)abc", r"\bword\b", r"\pL+", r"[[:alpha:]]+", r"[é-ü]", r"\Qa.b\E", r"a\1", r"(?s).", r"\x{1F600}"]


def _oracle_search(pattern, s):
    return celeval._re(pattern).search(s) is not None


def test_automata_match_the_oracle_on_the_corpus():
    for p in PATTERNS:
        dfa = regex.compile_regex(p)
        for s in STRINGS:
            assert dfa.search(s.encode("utf-8")) == _oracle_search(p, s), (p, s)


def test_patterns_outside_the_subset_are_refused():
    for p in OUTSIDE:
        with pytest.raises((regex.Unsupported, regex.Invalid)):
            regex.compile_regex(p)
    for p in ("a**", "(ab", "ab)", "[a-", "x{5,2}", "\\"):
        with pytest.raises(regex.Invalid):
            regex.compile_regex(p)


def _random_pattern(rng, depth=0):
    atoms = ["a", "b", "c", ".", r"\d", r"\w", r"\s", "[ab]", "[^a]", "[a-c]", "é", r"\.", "x"]
    r = rng.random()
    if depth > 2 or r < 0.45:
        node = str(rng.choice(atoms))
    elif r < 0.7:
        node = "".join(_random_pattern(rng, depth + 1) for _ in range(int(rng.integers(2, 4))))
    elif r < 0.85:
        node = "(" + "|".join(_random_pattern(rng, depth + 1) for _ in range(int(rng.integers(2, 4)))) + ")"
    else:
        node = "(?:" + _random_pattern(rng, depth + 1) + ")"
    q = rng.random()
    if q < 0.15:
        node = (node if len(node) == 1 or node[0] in "([\\" and _single(node) else "(" + node + ")") + str(rng.choice(["*", "+", "?", "{2}", "{1,3}", "{2,}", "*?"]))
    return node


def _single(node):
    return node.startswith("(") and node.endswith(")") or node.startswith("[") or (node.startswith("\\") and len(node) == 2)


def test_random_patterns_against_the_oracle():
    rng = np.random.default_rng(11)
    alphabet = list("abcx.1 _é\n")
    checked = 0
    for _ in range(300):
        body = _random_pattern(rng)
        p = ("^" if rng.random() < 0.3 else "") + body + ("$" if rng.random() < 0.3 else "")
        try:
            dfa = regex.compile_regex(p)
        except (regex.Unsupported, regex.Invalid):   # (the generator can stack two repetition operators: invalid RE2)
            continue
        for _ in range(40):
            s = "".join(str(rng.choice(alphabet)) for _ in range(int(rng.integers(0, 7))))
            assert dfa.search(s.encode("utf-8")) == _oracle_search(p, s), (p, s)
            checked += 1
    assert checked > 8000


def _docs(patterns):
    rules = [{"actions": ["p%d" % i], "roles": ["*"], "effect": "EFFECT_ALLOW",
              "condition": {"match": {"expr": "R.attr.s.matches(%s)" % _cel_string(p)}}} for i, p in enumerate(patterns)]
    rules.append({"actions": ["global_form"], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": 'matches(R.attr.s, "^a")'}}})
    return [{"apiVersion": API, "resourcePolicy": {"resource": "text", "version": "default", "rules": rules}}]


def _cel_string(s):
    return '"' + s.replace("\\", "\\\\").replace('"', '\\"') + '"'


def _run(make_evaluator, close):
    rt = rule_table_from_policies(policies_from_docs(_docs(PATTERNS)))
    lt = lower_rule_table(rt)
    assert not lt.unsupported, lt.unsupported
    actions = ["p%d" % i for i in range(len(PATTERNS))] + ["global_form"]
    inputs = [{"requestId": "q%d" % i, "actions": actions, "principal": {"id": "p", "roles": ["user"]},
               "resource": {"kind": "text", "id": "r%d" % i, "attr": {"s": s}}} for i, s in enumerate(STRINGS)]
    inputs.append({"requestId": "missing", "actions": actions, "principal": {"id": "p", "roles": ["user"]}, "resource": {"kind": "text", "id": "m", "attr": {}}})
    inputs.append({"requestId": "number", "actions": actions, "principal": {"id": "p", "roles": ["user"]}, "resource": {"kind": "text", "id": "n", "attr": {"s": 5.0}}})
    ev = make_evaluator(lt)
    try:
        outs, bad = ev.check(inputs, now_ns=NOW, allow_unsupported=True)
    finally:
        if close:
            ev.close()
    assert not bad
    orc = RuleTableOracle(rt)
    n_allow = 0
    for inp, have in zip(inputs, outs):
        want = orc.check(inp, EvalParams(now_ns=NOW))
        assert norm_actions(have) == norm_actions(want), (inp["resource"]["attr"], have["actions"], want["actions"])
        n_allow += sum(e["effect"] == "EFFECT_ALLOW" for e in want["actions"].values())
    assert 100 < n_allow < len(inputs) * len(actions) - 100

    lt2 = lower_rule_table(rule_table_from_policies(policies_from_docs(_docs([OUTSIDE[0]]))))
    assert lt2.unsupported
    ev2 = make_evaluator(lt2)
    try:
        _, bad2 = ev2.check([dict(inputs[0], actions=["p0"])], now_ns=NOW, allow_unsupported=True)
    finally:
        if close:
            ev2.close()
    assert bad2 == [0]


def test_matches_kernel_source_vs_oracle():
    from test_hostsim_golden import HostSimEvaluator
    _run(lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.gpu
def test_matches_on_gpu():
    _run(lambda lt: HipEvaluator(lt, Conf()), True)
