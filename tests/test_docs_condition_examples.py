"""The reference's PUBLISHED condition examples (docs/modules/policies/pages/conditions.adoc: per function family a request
fragment and a table of example expressions, mined by tools/make_golden_docs_conditions.py into
tests/golden/docs_condition_examples.json) through the oracle and through the device path (lowering + the kernel source on the
host simulator).  The documentation presents the examples as true statements about its test data; nothing in the reference runs
them, and seven do not hold against their own fragment (listed below with the reason) - the other 120 must evaluate to true in
the oracle, and on the device path decide ALLOW or be flagged UNSUPPORTED, never DENY."""
import pytest

from cerbos_amd.lower.celc import LoweringError
from helpers import CEL_EVAL_NOW_NS, load_json
from oracle import celeval
from oracle.check import EvalParams, _EvalContext
from test_hostsim_cel_kats import _decide

EXAMPLES = load_json("docs_condition_examples.json")["examples"]

# line of conditions.adoc -> why the statement is not one the engine can make true
NOT_TRUE = {
    376: 'cidr("192.168.0.0/8").isMask(): the address has bits outside its /8 prefix; the reference\'s own KAT (cel_eval/network.yaml) uses /24',
    380: "ip(x) == 4 compares an IP with an int (the row documents .family()): no such overload",
    426: "limits.design is 10 in the fragment: exactly one entry matches, exists_one is true and the statement says false",
    458: "math.bitNot(1) is -2 (the table repeats cel-go's README, which says -1)",
    492: r'"C:\path\to\dir" is not a CEL string literal (\p is no escape sequence)',
    554: "json.encode: outside the oracle's restated subset (DESIGN.md §2); the key order shown is not the sorted order a Go map marshals to",
    603: 'the fragment\'s lastAccessed is 10:00:20: getMinutes("UTC") is 0 (the KAT file cel_eval/timestamp_funcs.yaml has 10:05:20 and expects 5)',
}
# examples that are values, not statements
VALUES = {420: ["design", "communications", "product", "commercial", "design", "engineering"], 552: "department_marketing_1"}


def _ctx(ex):
    inp = dict(ex["request"])
    inp.setdefault("principal", {}).setdefault("id", "")
    inp.setdefault("resource", {}).setdefault("kind", "")
    return _EvalContext(EvalParams(now_ns=CEL_EVAL_NOW_NS), inp)


@pytest.mark.parametrize("ex", EXAMPLES, ids=lambda e: "L%d-%s" % (e["line"], e["function"].replace(" ", "_")[:24]))
def test_example_by_the_oracle(ex):
    if ex["line"] in NOT_TRUE:
        try:
            assert celeval.evaluate(ex["expr"], _ctx(ex)._env({}, {})) is not True, "listed as not true, but it is"
        except Exception:   # noqa: BLE001 - a syntax error, a missing overload or an unsupported function: not true either
            pass
        return
    got = celeval.evaluate(ex["expr"], _ctx(ex)._env({}, {}))
    if ex["line"] in VALUES:
        assert (list(got) if isinstance(got, (list, tuple)) else got) == VALUES[ex["line"]]
    else:
        assert got is True, ex["expr"]


def test_examples_on_the_device_path():
    """Each statement as the condition of an ALLOW rule, the fragment as the request."""
    decided = flagged = 0
    for ex in EXAMPLES:
        if ex["line"] in NOT_TRUE or ex["line"] in VALUES:
            continue
        case = {"name": "L%d" % ex["line"], "request": ex["request"]}
        try:
            got = _decide(case, [{"actions": ["doc"], "condition": {"match": {"expr": ex["expr"]}}}])
        except LoweringError:
            flagged += 1
            continue
        if got is None:
            flagged += 1
            continue
        assert got["doc"] is True, (ex["line"], ex["expr"])
        decided += 1
    assert decided + flagged == len(EXAMPLES) - len(NOT_TRUE) - len(VALUES)
    assert decided >= 95, (decided, flagged)
