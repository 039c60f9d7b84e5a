"""CPU tier: BatchingEvaluator (many small concurrent Check calls -> few device batches) must give every caller
exactly what a direct Check gives, in its own order - kernel source on the host simulator underneath."""
import threading

import numpy as np
import pytest

from cerbos_amd.coalesce import BatchingEvaluator
from cerbos_amd.engine import Conf
from cerbos_amd.lower.blob import lower_rule_table
from helpers import load_json, store_rule_table
from test_hostsim_golden import GLOBALS, HostSimEvaluator

NOW = 1_700_000_000_000_000_000


@pytest.fixture(scope="module")
def evaluator():
    return HostSimEvaluator(lower_rule_table(store_rule_table(), GLOBALS), Conf(globals_=GLOBALS))


def test_concurrent_calls_are_coalesced_and_answered_in_order(evaluator):
    inputs = [i for c in load_json("engine_cases.json") for i in c["inputs"]]
    rng = np.random.default_rng(5)
    calls = []
    for _ in range(120):
        k = int(rng.integers(1, 5))
        calls.append(([inputs[int(x)] for x in rng.integers(0, len(inputs), size=k)], bool(rng.integers(0, 2))))
    want = [evaluator.check(ins, now_ns=NOW, lenient_scope_search=len_, allow_unsupported=True) for ins, len_ in calls]
    be = BatchingEvaluator(evaluator, max_inputs=64, max_wait_s=0.01)
    got = [None] * len(calls)

    def worker(lo, hi):
        for j in range(lo, hi):
            ins, len_ = calls[j]
            got[j] = be.check(ins, now_ns=NOW, lenient_scope_search=len_, allow_unsupported=True)
    threads = [threading.Thread(target=worker, args=(a, a + 10)) for a in range(0, len(calls), 10)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    be.close()
    for (g_out, g_bad), (w_out, w_bad) in zip(got, want):
        assert g_out == w_out and list(g_bad) == list(w_bad)
    assert be.calls == len(calls) and be.batches < len(calls)      # something was actually batched


def test_errors_and_empty_calls(evaluator):
    be = BatchingEvaluator(evaluator, max_wait_s=0.0)
    assert be.check([]) == []
    assert be.check([], allow_unsupported=True) == ([], [])
    with pytest.raises(Exception):
        be.check([{"principal": {"id": "x"}}], now_ns=NOW)          # malformed input: no resource
    ok = be.check([load_json("engine_cases.json")[0]["inputs"][0]], now_ns=NOW, allow_unsupported=True)
    assert len(ok[0]) == 1                                            # the batcher survives a failed batch
    be.close()
    with pytest.raises(RuntimeError):
        be.check([load_json("engine_cases.json")[0]["inputs"][0]])


# ---- the same gathering on the bytes a server's handler holds: CheckResourcesRequests (RequestBatcher over check_requests_pb)
def _request_of(inputs, include_meta):
    return {"requestId": inputs[0].get("requestId", ""), "includeMeta": include_meta, "principal": inputs[0]["principal"],
            "resources": [{"actions": i["actions"], "resource": i["resource"]} for i in inputs]}


def test_request_bytes_are_gathered_and_answered_per_request(evaluator):
    from cerbos_amd import wire
    from cerbos_amd.coalesce import RequestBatcher
    cases = load_json("server_check_cases.json")
    rng = np.random.default_rng(9)
    picks = [cases[int(x)] for x in rng.integers(0, len(cases), size=90)]
    reqs = [wire.encode_check_resources_request(_request_of(c["inputs"], bool(k & 1))) for k, c in enumerate(picks)]
    want = [evaluator.check_requests_pb([r], now_ns=NOW) for r in reqs]
    rb = RequestBatcher(evaluator, max_requests=32, max_wait_s=0.01)
    got = [None] * len(reqs)

    def worker(lo, hi):
        for j in range(lo, hi):
            got[j] = rb.check_request(reqs[j], now_ns=NOW)
    threads = [threading.Thread(target=worker, args=(a, min(a + 9, len(reqs)))) for a in range(0, len(reqs), 9)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k, ((outs, flags, meta), (w_outs, w_flags, w_meta)) in enumerate(zip(got, want)):
        assert outs == w_outs[0] and list(flags) == list(w_flags) and meta == bool(w_meta[0]) == bool(k & 1)
        for raw, wnt in zip(outs, picks[k]["want"]):
            assert {a: e["effect"] for a, e in wire.decode_check_output(raw)["actions"].items()} == wnt["actions"]
    assert rb.calls == len(reqs) and rb.batches < len(reqs)
    with pytest.raises(Exception, match="malformed CheckResourcesRequest"):
        rb.check_request(reqs[0][:-2], now_ns=NOW)
    assert rb.check_request(reqs[1], now_ns=NOW)[0] == want[1][0][0]       # the batcher survives a failed batch
    rb.close()
    with pytest.raises(RuntimeError):
        rb.check_request(reqs[0])


def test_a_coalesced_batch_keeps_a_trail_per_request(evaluator):
    """Decision logs on: every handler gets the EffectivePolicies of ITS request out of the batch its request rode in."""
    from cerbos_amd import wire
    from cerbos_amd.coalesce import RequestBatcher
    cases = [c for c in load_json("server_check_cases.json")]
    reqs = [wire.encode_check_resources_request(_request_of(c["inputs"], False)) for c in cases]
    alone = [evaluator.check_requests_pb([r], now_ns=NOW, audit_trail=True) for r in reqs]
    rb = RequestBatcher(evaluator, max_requests=64, max_wait_s=0.02, audit_trail=True)
    got = [None] * len(reqs)

    def worker(j):
        got[j] = rb.check_request(reqs[j], now_ns=NOW)
    threads = [threading.Thread(target=worker, args=(j,)) for j in range(len(reqs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    rb.close()
    assert rb.batches < len(reqs)
    for (outs, flags, meta, trail), (w_outs, w_flags, _, w_trail) in zip(got, alone):
        assert outs == w_outs[0] and list(flags) == list(w_flags) and trail == w_trail[0]
    assert len({tuple(g[3]) for g in got}) > 1 and all(g[3] for g in got)
