"""Cost of the trace pass (cbh_trace_batch) next to the decision pass, one-shot path, on the GPU box.
   python tools/gpu_trace_bench.py [C3|C5] [n_requests]  -> one JSON line
Both calls include the PCIe copies (pageable host arrays); the trace pass decides the batch again with programs that keep
expression identities and logs one record per failing expression."""
import json
import sys
import time

sys.path.insert(0, ".")
from cerbos_amd import capi, workloads  # noqa: E402
from cerbos_amd.flatten import Flattener  # noqa: E402
from cerbos_amd.lower.blob import lower_rule_table  # noqa: E402
from cerbos_amd.policy.loader import policies_from_docs  # noqa: E402
from cerbos_amd.ruletable.build import rule_table_from_policies  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
pol, reqs = {"C3": (workloads.c3_policies, workloads.c3_requests), "C5": (workloads.c5_policies, workloads.c5_requests)}[name]
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
batch = Flattener(lt).flatten(reqs(n_requests=n).to_inputs())
capi.init(0)
table = capi.Table(lt.blob)
NOW = 1_700_000_000_000_000_000


def best(fn, k=5):
    fn()
    t = 1e9
    for _ in range(k):
        t0 = time.perf_counter()
        out = fn()
        t = min(t, time.perf_counter() - t0)
    return t, out


t_check, res = best(lambda: table.check(batch, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES, device_order=True))
t_trace, (tres, rec) = best(lambda: table.trace(batch, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES))
assert (tres.effect == res.effect).all()
marked = int((res.status == capi.ST_CEL_ERROR).sum())
print(json.dumps({"workload": name, "requests": n, "tuples": int(batch.n_tuples), "tuples_marked_cel_error": marked,
                  "trace_records": int(len(rec)), "check_batch_ms": t_check * 1e3, "trace_batch_ms": t_trace * 1e3,
                  "check_decisions_per_s": batch.n_tuples / t_check, "trace_decisions_per_s": batch.n_tuples / t_trace,
                  "note": "one-shot calls on pageable arrays, PCIe copies included; trace = whole batch, the product traces only the marked inputs"}))
