"""The reference's quickstart, end to end (docs/modules/ROOT/pages/quickstart.adoc, mined by tools/make_golden_quickstart.py into
tests/golden/quickstart.json): one CheckResourcesRequest against three stages of a policy directory - empty; a derived-roles file
and a resource policy; the policy with one more rule - and the CheckResourcesResponse the documentation publishes after each.
Answered by the policy-level oracle and by the bytes path of the product on the simulator: serialized CheckResourcesRequest ->
libcerbos_ingest.so -> the kernel source -> serialized CheckResourcesResponse."""
import pytest

import hostsim_api
from cerbos_amd import capi, wire
from cerbos_amd.ingest import IngestTable
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import load_json
from oracle.check import EvalParams, RuleTableOracle

QS = load_json("quickstart.json")
NOW = 1_700_000_000_000_000_000


def _inputs(req):
    return [{"requestId": req["requestId"], "principal": req["principal"], "resource": e["resource"], "actions": e["actions"]}
            for e in req["resources"]]


@pytest.mark.parametrize("stage", range(len(QS["stages"])), ids=lambda i: "stage%d" % i)
def test_quickstart_stage(stage):
    st = QS["stages"][stage]
    want = st["response"]
    assert want["requestId"] == QS["request"]["requestId"] and len(want["results"]) == len(QS["request"]["resources"])
    rt = rule_table_from_policies(policies_from_docs(st["policies"]))
    # the oracle, from the policies
    orc = RuleTableOracle(rt)
    for inp, w in zip(_inputs(QS["request"]), want["results"]):
        have = orc.check(inp, EvalParams(now_ns=NOW))
        assert {a: e["effect"] for a, e in have["actions"].items()} == w["actions"], (stage, inp["resource"]["id"])
    # the product: request bytes in, response bytes out
    lt = lower_rule_table(rt)
    it = IngestTable(lt.blob)
    req = wire.encode_check_resources_request(QS["request"])
    batch = it.flatten_request_pb(req)
    res = hostsim_api.check(lt, batch, NOW, capi.F_WANT_DERIVED_ROLES, device_order=True)
    raw, flags = it.assemble_response_pb(batch, res, req)
    resp = wire.decode_check_resources_response(raw)
    assert resp["requestId"] == want["requestId"] and len(resp["results"]) == len(want["results"])
    for have, w, f in zip(resp["results"], want["results"], flags):
        assert not f & 1
        assert (have["resource"]["id"], have["resource"]["kind"]) == (w["resource"]["id"], w["resource"]["kind"])
        assert have["actions"] == w["actions"], stage
