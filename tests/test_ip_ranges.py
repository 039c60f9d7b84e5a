"""inIPAddrRange (internal/conditions/cerbos_lib.go:513-526: net.ParseIP + net.ParseCIDR + IPNet.Contains) on the device
path for IPv4 and IPv6 against the oracle: compressed and full IPv6 forms, embedded IPv4, IPv4-mapped addresses (an IPv4
address to Go), prefix lengths 0..128, addresses that do not parse (an evaluation error), mixed families (not contained).
An IPv4-mapped IPv6 NETWORK is the one form the device flags for the caller's engine.
CPU tier: the kernel source on the host simulator; GPU tier: the kernel."""
import ipaddress

import numpy as np
import pytest

from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle

API = "api.cerbos.dev/v1"
NOW = 1_700_000_000_000_000_000
CONDS = {"kat6": 'P.attr.ip.inIPAddrRange("2001:db8::/48")', "kat4": 'P.attr.ip.inIPAddrRange("192.168.0.0/24")',
         "dyn": "P.attr.ip.inIPAddrRange(R.attr.cidr)", "not_dyn": "!P.attr.ip.inIPAddrRange(R.attr.cidr)"}


def _docs():
    return [{"apiVersion": API, "resourcePolicy": {"resource": "net", "version": "default", "rules": [
        {"actions": [n], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": e}}} for n, e in CONDS.items()]}}]


def _v6_text(rng, value):
    a = ipaddress.IPv6Address(value)
    r = rng.random()
    if r < 0.4:
        return a.compressed
    if r < 0.6:
        return a.exploded
    if r < 0.75:
        return a.exploded.upper()
    groups = a.exploded.split(":")
    if r < 0.9:   # embedded IPv4 in the last 32 bits
        return ":".join(groups[:6]) + ":" + str(ipaddress.IPv4Address(value & 0xFFFFFFFF))
    return ":".join(g.lstrip("0") or "0" for g in groups)


def _cases(rng, n):
    out = []
    for _ in range(n):
        fam = 6 if rng.random() < 0.6 else 4
        width = 128 if fam == 6 else 32
        bits = int(rng.choice([0, 1, 8, 24, 31, 32, 47, 48, 63, 64, 65, 96, 127, 128])) if fam == 6 else int(rng.integers(0, 33))
        bits = min(bits, width)
        base = int.from_bytes(rng.bytes(16 if fam == 6 else 4), "big")
        if fam == 6 and rng.random() < 0.3:
            base = (0x20010DB8 << 96) | (base & ((1 << 96) - 1))
        net = (base >> (width - bits)) << (width - bits) if bits else 0
        inside = rng.random() < 0.5
        host = net | (int.from_bytes(rng.bytes(16), "big") & ((1 << (width - bits)) - 1)) if inside else int.from_bytes(rng.bytes(16 if fam == 6 else 4), "big")
        host &= (1 << width) - 1
        ip = _v6_text(rng, host) if fam == 6 else str(ipaddress.IPv4Address(host))
        cidr = "%s/%d" % (_v6_text(rng, base) if fam == 6 else str(ipaddress.IPv4Address(base)), bits)
        r = rng.random()
        if r < 0.08:      # the other family on one side
            ip = "10.1.2.3" if fam == 6 else "2001:db8::1"
        elif r < 0.14 and fam == 4:   # an IPv4-mapped address is the IPv4 address
            ip = "::ffff:" + ip
        elif r < 0.22:    # does not parse
            ip = str(rng.choice(["", "1.2.3", "1.2.3.4.5", "01.2.3.4", "256.1.1.1", "1::2::3", "12345::", ":1", "1:2:3:4:5:6:7:8:9", "fe80::1%eth0",
                                 "1:2:3:4:5:6:7::8", "::1.2.3", "g::1", "1:2:3:4:5:6:7:"]))
        elif r < 0.28:
            cidr = str(rng.choice(["10.0.0.0", "10.0.0.0/33", "2001:db8::/129", "10.0.0.0/", "10.0.0.0/x", "10.0.0.0/255.0.0.0", "::/-1", "/8"]))
        out.append((ip, cidr))
    out += [("2001:db8:0:1::5", "2001:db8::/48"), ("2001:db9::5", "2001:db8::/48"), ("::", "::/0"), ("1.2.3.4", "0.0.0.0/0"),
            ("::ffff:10.0.0.1", "10.0.0.0/8"), ("10.0.0.1", "10.0.0.0/08"), ("10.0.0.1", "10.0.0.0/008"), ("::1", "::1/128"), ("::2", "::1/128"),
            ("192.168.0.77", "192.168.0.0/16"), ("::ffff:192.168.0.9", "0.0.0.0/0"), ("192.168.1.77", "192.168.0.0/16")]
    return out


def _run(make_evaluator, close):
    rng = np.random.default_rng(6)
    rt = rule_table_from_policies(policies_from_docs(_docs()))
    lt = lower_rule_table(rt)
    assert not lt.unsupported, lt.unsupported
    cases = _cases(rng, 500)
    inputs = [{"requestId": "q%d" % i, "actions": list(CONDS), "principal": {"id": "p", "roles": ["user"], "attr": {"ip": ip}},
               "resource": {"kind": "net", "id": "r%d" % i, "attr": {"cidr": cidr}}} for i, (ip, cidr) in enumerate(cases)]
    mapped_net = dict(inputs[0], requestId="mapped", actions=["dyn"], resource={"kind": "net", "id": "m", "attr": {"cidr": "::ffff:10.0.0.0/104"}})
    ev = make_evaluator(lt)
    try:
        outs, bad = ev.check(inputs, now_ns=NOW, allow_unsupported=True)
        _, bad_mapped = ev.check([mapped_net], now_ns=NOW, allow_unsupported=True)
    finally:
        if close:
            ev.close()
    assert not bad and bad_mapped == [0]
    orc = RuleTableOracle(rt)
    allowed = dict.fromkeys(CONDS, 0)
    errors = 0
    for inp, have in zip(inputs, outs):
        want = orc.check(inp, EvalParams(now_ns=NOW))
        assert norm_actions(have) == norm_actions(want), (inp["principal"]["attr"], inp["resource"]["attr"], have["actions"], want["actions"])
        errors += bool(want.get("evaluationErrors"))
        for a, e in want["actions"].items():
            allowed[a] += e["effect"] == "EFFECT_ALLOW"
    assert all(v > 0 for v in allowed.values()) and allowed["dyn"] > 100 and allowed["not_dyn"] > 100 and errors > 40, (allowed, errors)


def test_ip_ranges_kernel_source_vs_oracle():
    from test_hostsim_golden import HostSimEvaluator
    _run(lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.gpu
def test_ip_ranges_on_gpu():
    _run(lambda lt: HipEvaluator(lt, Conf()), True)
