"""Device CEL added in round 3, against the oracle over generated inputs:

* string WINDOWS as ropes - substring / charAt / trim - and replace() (cel-go ext/strings.go) on strings the request
  supplies: code-point indices on non-ASCII text, out-of-range indices (a CEL error), all of Go's Unicode spaces;
* cel-go ext.Network on request strings: isIP, ip().family / isLoopback / isUnspecified / isLinkLocalUnicast /
  isLinkLocalMulticast / isGlobalUnicast, ip.isCanonical, cidr(constant).containsIP - IPv4 and IPv6 texts, compressed and
  expanded forms, upper-case hex, zones and IPv4-mapped forms (refused), things that are no address at all;
* hierarchy(a).commonAncestors(hierarchy(b)) == hierarchy(c) over three request strings.

Per action: effect; per request: whether evaluation errors were recorded.  CPU tier: the kernel source on the host
simulator; GPU tier: the kernel."""
import numpy as np
import pytest

from cerbos_amd import capi
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000
STR_CONDS = {
    "sub1": 'R.attr.s.substring(1) == P.attr.n', "sub2": 'R.attr.s.substring(1, 3) == "bé"', "sub_var": 'R.attr.s.substring(int(P.attr.i)) == P.attr.n',
    "sub_range": 'R.attr.s.substring(2, int(P.attr.j)) != ""', "char": 'R.attr.s.charAt(1) == "b"', "char_var": 'R.attr.s.charAt(int(P.attr.i)) == P.attr.n', "char_dbl": 'R.attr.s.charAt(P.attr.i) == "a"',
    "char_end": 'R.attr.s.charAt(size(R.attr.s)) == ""', "trim": 'R.attr.s.trim() == P.attr.n', "trim_size": 'size(R.attr.s.trim()) < size(R.attr.s)',
    "rep": 'R.attr.s.replace("ab", "X") == P.attr.n', "rep_in": 'R.attr.s.replace("a", "") in ["b", "bb", ""]',
    "rep_starts": 'R.attr.s.replace("b", "日本").startsWith("a日")', "sub_contains": 'R.attr.s.substring(1).contains(P.attr.n)',
    "cat": '(R.attr.s.substring(0, 1) + R.attr.s.trim()) == P.attr.n',
}
NET_CONDS = {
    "isip": "isIP(R.attr.a)", "isip4": "isIP(R.attr.a, 4)", "isip6": "isIP(R.attr.a, 6)", "fam4": "ip(R.attr.a).family() == 4",
    "loop": "ip(R.attr.a).isLoopback()", "unspec": "ip(R.attr.a).isUnspecified()", "llu": "ip(R.attr.a).isLinkLocalUnicast()",
    "llm": "ip(R.attr.a).isLinkLocalMulticast()", "glob": "ip(R.attr.a).isGlobalUnicast()", "canon": "ip.isCanonical(R.attr.a)",
    "in4": 'cidr("192.168.0.0/16").containsIP(R.attr.a)', "in6": 'cidr("2001:db8::/32").containsIP(ip(R.attr.a))',
    "in_unmasked": 'cidr("10.1.2.3/8").containsIP(R.attr.a)',
}
HIER_CONDS = {
    "common": 'hierarchy(R.attr.x).commonAncestors(hierarchy(P.attr.y)) == hierarchy(P.attr.z)',
    "common_ne": 'hierarchy(R.attr.x).commonAncestors(hierarchy(P.attr.y)) != hierarchy("a.b")',
    "common_rev": 'hierarchy("a") == hierarchy(R.attr.x).commonAncestors(hierarchy(P.attr.y))',
}
ADDRS = ["127.0.0.1", "127.255.0.3", "0.0.0.0", "255.255.255.255", "192.168.4.7", "192.169.0.1", "10.200.1.1", "169.254.9.9", "224.0.0.5",
         "224.0.1.5", "239.1.1.1", "8.8.8.8", "01.2.3.4", "1.2.3", "1.2.3.4.5", "256.1.1.1", "1.2.3.4 ", "", "::", "::1", "::2", "fe80::1",
         "febf::9", "fec0::1", "ff02::1", "ff05::2", "ff12::3", "2001:db8::1", "2001:DB8::1", "2001:db8:0:0:0:0:0:1", "2001:db8:0:0:1:0:0:1",
         "2001:db8::1:0:0:1", "2001:0db8::1", "2001:db9::", "::ffff:1.2.3.4", "::ffff:102:304", "fe80::1%eth0", "1:2:3:4:5:6:7:8",
         "1:2:3:4:5:6:7::", "1:0:0:4:0:0:0:8", "1::4:0:0:0:8", "1:0:0:4::8", "::1.2.3.4", "64:ff9b::1.2.3.4", "g::1", "1:::2", "nope", ":1", "1:"]
SEGS = ["a", "b", "c", "a.b", "a.b.c", "a.b.d", "a.x", "", "a.", ".a", "a..b", "b.a", "a.b.c.d"]


def _docs(conds, kind):
    return [{"apiVersion": API, "resourcePolicy": {"resource": kind, "version": "default", "rules": [
        {"actions": [n], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": e}}} for n, e in conds.items()]}}]


def _compare(make_evaluator, close, conds, kind, inputs, min_discriminating):
    rt = rule_table_from_policies(policies_from_docs(_docs(conds, kind)))
    lt = lower_rule_table(rt)
    assert not lt.unsupported, lt.unsupported
    ev = make_evaluator(lt)
    try:
        batch = Flattener(lt).flatten(inputs)
        res = ev.table.check(batch, now_ns=NOW, flags=0)
        outs, bad = ev.assemble(inputs, batch, res, "default", allow_unsupported=True)
    finally:
        if close:
            ev.close()
    assert not bad
    orc = RuleTableOracle(rt)
    allowed, denied, t = dict.fromkeys(conds, 0), dict.fromkeys(conds, 0), 0
    for inp, have in zip(inputs, outs):
        want = orc.check(inp, EvalParams(now_ns=NOW))
        assert norm_actions(have) == norm_actions(want), (inp["resource"]["attr"], inp["principal"].get("attr"), have["actions"], want["actions"])
        na = len(inp["actions"])
        assert bool((res.status[t:t + na] == capi.ST_CEL_ERROR).any()) == bool(want.get("evaluationErrors")), (inp, want.get("evaluationErrors"))
        t += na
        for a, e in want["actions"].items():
            allowed[a] += e["effect"] == "EFFECT_ALLOW"
            denied[a] += e["effect"] != "EFFECT_ALLOW"
    assert sum(allowed[a] > 0 and denied[a] > 0 for a in conds) >= min_discriminating, (allowed, denied)


def _string_inputs():
    rng = np.random.default_rng(303)
    alphabet = list("abAB ") + ["é", "日", "\t", " ", " ", "　", "\n"]
    inputs = []
    for i in range(500):
        s = "".join(str(rng.choice(alphabet)) for _ in range(int(rng.integers(0, 8))))
        r = rng.random()
        if r < 0.25:
            n = s.strip(" \t\n  　")
        elif r < 0.5:
            n = s[1:]
        elif r < 0.65:
            n = s.replace("ab", "X")
        elif r < 0.8:
            n = s[:1] + s.strip(" \t\n  　")
        elif r < 0.9 and s:
            n = s[min(len(s) - 1, 1)]
        else:
            n = "".join(str(rng.choice(alphabet)) for _ in range(int(rng.integers(0, 3))))
        pattr = {"n": n, "i": float(rng.integers(-1, 9)), "j": float(rng.integers(0, 9))}
        if rng.random() < 0.1:
            pattr["i"] = 1.5
        inputs.append({"requestId": "q%d" % i, "actions": list(STR_CONDS), "principal": {"id": "p", "roles": ["user"], "attr": pattr},
                       "resource": {"kind": "text", "id": "r%d" % i, "attr": {"s": s}}})
    for k, s in enumerate(("abé", "abébé", "ab", "aab", "abab", "  ab  ", "b")):
        inputs.append(dict(inputs[0], requestId="fixed%d" % k, resource={"kind": "text", "id": "f%d" % k, "attr": {"s": s}},
                           principal={"id": "p", "roles": ["user"], "attr": {"n": s[1:], "i": 1.0, "j": 3.0}}))
    inputs.append(dict(inputs[0], requestId="num", resource={"kind": "text", "id": "x", "attr": {"s": 7.0}}))
    inputs.append(dict(inputs[0], requestId="missing", resource={"kind": "text", "id": "y", "attr": {}}))
    return inputs


def _net_inputs():
    return [{"requestId": "n%d" % i, "actions": list(NET_CONDS), "principal": {"id": "p", "roles": ["user"]},
             "resource": {"kind": "host", "id": "h%d" % i, "attr": ({"a": a} if a is not None else {})}}
            for i, a in enumerate(ADDRS + [None, 12.0, True])]


def _hier_inputs():
    rng = np.random.default_rng(404)
    out = []
    for i in range(300):
        x, y = str(rng.choice(SEGS)), str(rng.choice(SEGS))
        z = str(rng.choice(SEGS)) if rng.random() < 0.5 else ".".join(p for p, q in zip(x.split("."), y.split(".")) if p == q)
        out.append({"requestId": "h%d" % i, "actions": list(HIER_CONDS), "principal": {"id": "p", "roles": ["user"], "attr": {"y": y, "z": z}},
                    "resource": {"kind": "tree", "id": "t%d" % i, "attr": {"x": x}}})
    out.append(dict(out[0], requestId="hnum", resource={"kind": "tree", "id": "n", "attr": {"x": 3.0}}))
    out.append(dict(out[0], requestId="hmiss", resource={"kind": "tree", "id": "m", "attr": {}}))
    return out


def _all(make, close):
    _compare(make, close, STR_CONDS, "text", _string_inputs(), len(STR_CONDS) - 3)
    _compare(make, close, NET_CONDS, "host", _net_inputs(), len(NET_CONDS) - 1)
    _compare(make, close, HIER_CONDS, "tree", _hier_inputs(), len(HIER_CONDS))


def test_kernel_source_vs_oracle():
    from test_hostsim_golden import HostSimEvaluator
    _all(lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.gpu
def test_on_gpu():
    _all(lambda lt: HipEvaluator(lt, Conf()), True)
