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
