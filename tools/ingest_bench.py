"""Host ingest throughput: serialized CheckInput bytes -> cbh_batch by libcerbos_ingest.so, next to the Python
flattener on the same requests.  python tools/ingest_bench.py [C2|C3|C5] [n_requests]"""
import sys
import time

sys.path.insert(0, ".")
from cerbos_amd import wire, workloads  # noqa: E402
from cerbos_amd.flatten import Flattener  # noqa: E402
from cerbos_amd.ingest import IngestTable  # noqa: E402
from cerbos_amd.lower.blob import lower_rule_table  # noqa: E402
from cerbos_amd.policy.loader import policies_from_docs  # noqa: E402
from cerbos_amd.ruletable.build import rule_table_from_policies  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
pol, reqs = {"C2": (workloads.c2_policies, workloads.c2_requests), "C3": (workloads.c3_policies, workloads.c3_requests),
             "C5": (workloads.c5_policies, workloads.c5_requests)}[name]
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
inputs = reqs(n_requests=n).to_inputs()
data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
it = IngestTable(lt.blob)
it.flatten_pb(data, off)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    b = it.flatten_pb(data, off)
    best = min(best, time.perf_counter() - t0)
t0 = time.perf_counter()
Flattener(lt).flatten(inputs)
tpy = time.perf_counter() - t0
print("%s: %d requests, %d tuples, %.1f MB of wire bytes (%.0f B/request)" % (name, n, b.n_tuples, data.size / 1e6, data.size / n))
print("C++ ingest (1 thread, incl. copy-out to numpy): %.1f ms = %.2f M requests/s = %.2f M decisions/s, %.0f MB/s"
      % (best * 1e3, n / best / 1e6, b.n_tuples / best / 1e6, data.size / best / 1e6))
print("Python flattener: %.1f ms = %.3f M requests/s (%.0fx slower)" % (tpy * 1e3, n / tpy / 1e6, tpy / best))

# the table is immutable: independent slices of the input flatten concurrently (ctypes drops the GIL), each
# into its own cbh_batch - what a Go caller does with one goroutine per slice
import os  # noqa: E402
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

import numpy as np  # noqa: E402

for threads in (2, 4, 8, 16, 32):
    if threads > (os.cpu_count() or 1):
        break
    cuts = np.linspace(0, n, threads + 1).astype(int)
    parts = [(data[int(off[a]):int(off[b])], off[a:b + 1] - off[a]) for a, b in zip(cuts[:-1], cuts[1:])]
    with ThreadPoolExecutor(threads) as ex:
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            list(ex.map(lambda p: it.flatten_pb(p[0], p[1]), parts))
            best = min(best, time.perf_counter() - t0)
    print("%2d threads: %.1f ms = %.2f M requests/s" % (threads, best * 1e3, n / best / 1e6))

# the same, inside ONE call (cbi_flatten_pb_mt): slices flattened concurrently and merged into one batch
for threads in (1, 2, 4, 8, 16, 32):
    if threads > (os.cpu_count() or 1):
        break
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        it.flatten_pb(data, off, threads=threads)
        best = min(best, time.perf_counter() - t0)
    print("in-call %2d threads: %.1f ms = %.2f M requests/s" % (threads, best * 1e3, n / best / 1e6))
