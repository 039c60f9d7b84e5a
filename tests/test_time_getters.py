"""Timestamp / duration getters on the device path (cel-go timestamp.go / duration.go: getFullYear ... getMilliseconds,
optional fixed-offset zone) against oracle/celeval.py over random instants: leap days, year and day boundaries that a
zone offset moves across, pre-1970 instants, negative durations.  IANA zone names: their offsets are lowered into the table
(1900 .. 2100; checked against zoneinfo through the oracle, daylight-saving edges included).  An unknown zone must be
flagged UNSUPPORTED, never answered.  CPU tier: the kernel source on the host simulator; GPU tier: the kernel."""
import datetime

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
TS = "timestamp(R.attr.t)"
DUR = "duration(R.attr.d)"
CONDS = {
    "year": TS + '.getFullYear("+14:00") == 2024', "year_utc": TS + ".getFullYear() >= 2000",
    "month": TS + '.getMonth("+05:30") >= 6', "month_utc": TS + ".getMonth() == 0",
    "doy": TS + ".getDayOfYear() > 180", "doy_tz": TS + '.getDayOfYear("-12:00") == 0',
    "dom": TS + ".getDayOfMonth() == 0", "date": TS + '.getDate("-12:00") == 1', "date_utc": TS + '.getDate("UTC") >= 29',
    "dow": TS + ".getDayOfWeek() == 0", "dow_tz": TS + '.getDayOfWeek("-05:00") >= 5',
    "hours": TS + '.getHours("-05:00") < 12', "hours_utc": TS + ".getHours() == 23",
    "minutes": TS + '.getMinutes("+05:30") < 30', "seconds": TS + ".getSeconds() >= 30", "millis": TS + ".getMilliseconds() < 500",
    "d_hours": DUR + ".getHours() >= 2", "d_minutes": DUR + ".getMinutes() == -90", "d_seconds": DUR + ".getSeconds() < 0",
    "d_millis": DUR + ".getMilliseconds() > 3600000",
    "d_no_overload": DUR + ".getFullYear() == 1",       # no such overload: a CEL error, not an answer
    "unknown_zone": TS + '.getHours("Mars/Olympus_Mons") == 1',   # not in the zone database: UNSUPPORTED at lowering
}
# IANA zones: their offsets over 1900 .. 2100 are lowered into the table (celc.py _named_zone_table); instants outside are flagged
NAMED = {
    "nz_hours": TS + '.getHours("Pacific/Auckland") == 1', "nz_month": TS + '.getMonth("NZ") == 3',
    "ny_dow": TS + '.getDayOfWeek("America/New_York") >= 5', "lon_date": TS + '.getDate("Europe/London") == 1',
    "kolkata_min": TS + '.getMinutes("Asia/Kolkata") < 30', "lord_howe_min": TS + '.getMinutes("Australia/Lord_Howe") >= 30',
    "ny_doy": TS + '.getDayOfYear("America/New_York") == 0', "utc_name": TS + '.getHours("Etc/UTC") == 23',
}
CONDS.update(NAMED)
SUPPORTED = [k for k in CONDS if k != "unknown_zone" and k not in NAMED]


def _docs(names):
    return [{"apiVersion": API, "resourcePolicy": {"resource": "clock", "version": "default", "rules": [
        {"actions": [n], "roles": ["*"], "effect": "EFFECT_ALLOW", "condition": {"match": {"expr": CONDS[n]}}} for n in names]}}]


def _instants(rng, n):
    out = []
    for i in range(n):
        r = rng.random()
        if r < 0.25:     # around a year boundary
            base = datetime.datetime(int(rng.integers(1701, 2200)), 1, 1) + datetime.timedelta(seconds=int(rng.integers(-50_000, 50_000)))
        elif r < 0.4:    # leap days and the days after
            base = datetime.datetime(int(rng.choice([1904, 1972, 2000, 2024, 2096])), 2, 29) + datetime.timedelta(seconds=int(rng.integers(0, 200_000)))
        else:
            base = datetime.datetime(1700, 1, 1) + datetime.timedelta(seconds=int(rng.integers(0, 500 * 365 * 86_400)))
        ms = int(rng.integers(0, 1000))
        off = int(rng.choice([0, 0, -300, 330, 840, -720]))
        sign = "+" if off >= 0 else "-"
        tz = "Z" if off == 0 and rng.random() < 0.5 else "%s%02d:%02d" % (sign, abs(off) // 60, abs(off) % 60)
        out.append("%04d-%02d-%02dT%02d:%02d:%02d.%03d%s" % (base.year, base.month, base.day, base.hour, base.minute, base.second, ms, tz))
    return out


def _durations(rng, n):
    units = ["h", "m", "s", "ms"]
    return ["%s%d%s%d%s" % ("-" if rng.random() < 0.3 else "", int(rng.integers(0, 50)), units[int(rng.integers(0, 2))],
                             int(rng.integers(0, 5000)), units[int(rng.integers(2, 4))]) for _ in range(n)] + ["-1h30m", "1h0m0.001s"]


def _run(make_evaluator, close):
    rng = np.random.default_rng(77)
    rt = rule_table_from_policies(policies_from_docs(_docs(SUPPORTED)))
    lt = lower_rule_table(rt)
    assert not lt.unsupported, lt.unsupported
    ts, ds = _instants(rng, 300), _durations(rng, 298)
    inputs = [{"requestId": "q%d" % i, "actions": SUPPORTED, "principal": {"id": "p", "roles": ["user"]},
               "resource": {"kind": "clock", "id": "r%d" % i, "attr": {"t": ts[i], "d": ds[i]}}} for i in range(300)]
    ev = make_evaluator(lt)
    try:
        outs, bad = ev.check(inputs, now_ns=NOW, allow_unsupported=True)
    finally:
        if close:
            ev.close()
    assert not bad
    orc = RuleTableOracle(rt)
    allowed = dict.fromkeys(SUPPORTED, 0)
    for inp, have in zip(inputs, outs):
        want = orc.check(inp, EvalParams(now_ns=NOW))
        assert norm_actions(have) == norm_actions(want), (inp["resource"]["attr"], have["actions"], want["actions"])
        for a, e in want["actions"].items():
            allowed[a] += e["effect"] == "EFFECT_ALLOW"
    assert all(v > 0 for k, v in allowed.items() if k not in ("d_no_overload", "d_minutes", "doy_tz")), allowed   # the conditions discriminate
    assert allowed["d_no_overload"] == 0

    # a zone the database does not know: flagged, not answered
    lt2 = lower_rule_table(rule_table_from_policies(policies_from_docs(_docs(["unknown_zone"]))))
    assert lt2.unsupported
    ev2 = make_evaluator(lt2)
    try:
        _, bad2 = ev2.check([dict(inputs[0], actions=["unknown_zone"])], now_ns=NOW, allow_unsupported=True)
    finally:
        if close:
            ev2.close()
    assert bad2 == [0]

    # IANA zones: instants of 1900 .. 2100 answer as the oracle (zoneinfo) does - the seconds around daylight-saving changes of
    # both hemispheres included -, instants outside are flagged
    names = list(NAMED)
    rt3 = rule_table_from_policies(policies_from_docs(_docs(names)))
    lt3 = lower_rule_table(rt3)
    assert not lt3.unsupported, lt3.unsupported
    edges = []
    for iso in ("2021-04-03T14:00:00", "2021-09-25T14:00:00", "2021-03-14T07:00:00", "2021-11-07T06:00:00", "2021-03-28T01:00:00",
                "2021-10-31T01:00:00", "1974-01-06T07:00:00", "2037-12-31T23:59:59", "2099-12-31T12:59:59", "1900-01-01T00:00:00",
                "2021-04-03T15:30:00", "2021-10-02T15:30:00", "1945-08-14T23:00:00"):
        base = datetime.datetime.fromisoformat(iso)
        for d in (-1, 0, 1, 1799, 3600):
            edges.append((base + datetime.timedelta(seconds=d)).strftime("%Y-%m-%dT%H:%M:%SZ"))
    ts3 = edges + _instants(rng, 250) + ["1899-12-31T23:59:59Z", "2100-01-01T00:00:00Z", "1850-06-01T00:00:00Z", "2150-06-01T00:00:00Z"]
    in3 = [{"requestId": "z%d" % i, "actions": names, "principal": {"id": "p", "roles": ["user"]},
            "resource": {"kind": "clock", "id": "z%d" % i, "attr": {"t": t}}} for i, t in enumerate(ts3)]
    ev3 = make_evaluator(lt3)
    try:
        outs3, bad3 = ev3.check(in3, now_ns=NOW, allow_unsupported=True)
    finally:
        if close:
            ev3.close()
    from oracle import celeval
    orc3 = RuleTableOracle(rt3)
    allowed3, inside = dict.fromkeys(names, 0), 0
    for i, (inp, have) in enumerate(zip(in3, outs3)):
        sec = celeval.parse_timestamp(inp["resource"]["attr"]["t"]).ns // 1_000_000_000
        if not (-2208988800 <= sec < 4102444800):
            assert i in bad3, inp["resource"]["attr"]
            continue
        assert i not in bad3, inp["resource"]["attr"]
        inside += 1
        want = orc3.check(inp, EvalParams(now_ns=NOW))
        assert norm_actions(have) == norm_actions(want), (inp["resource"]["attr"], have["actions"], want["actions"])
        for a, e in want["actions"].items():
            allowed3[a] += e["effect"] == "EFFECT_ALLOW"
    assert inside > 150 and all(v > 0 for v in allowed3.values()), (inside, allowed3)


def test_getters_kernel_source_vs_oracle():
    from test_hostsim_golden import HostSimEvaluator
    _run(lambda lt: HostSimEvaluator(lt, Conf()), False)


@pytest.mark.gpu
def test_getters_on_gpu():
    _run(lambda lt: HipEvaluator(lt, Conf()), True)
