"""Wire-inclusive end-to-end rate on one GPU box: serialized CheckInput bytes -> libcerbos_ingest.so (flatten) ->
cbh_check_batch (H2D, kernels, D2H over PCIe) -> cbi_assemble_pb (serialized CheckOutput), from T host threads
that each work through slices of the request stream - the shape of a Go server with one goroutine per slice.
Not the bench.py metric (that one is the HBM-resident rate); this is the number DESIGN.md quotes next to it.

    python tools/e2e_bench.py [C2|C3] [n_requests] [slice_requests] [seconds] [threads,threads,...] [inner]

`inner` > 1 uses cbi_flatten_pb_mt / cbi_assemble_pb_mt with that many threads inside each call (fewer, larger
slices per Python thread).
"""
import ctypes as C
import json
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from cerbos_amd import capi, ingest, wire, workloads  # noqa: E402
from cerbos_amd.lower.blob import lower_rule_table  # noqa: E402
from cerbos_amd.policy.loader import policies_from_docs  # noqa: E402
from cerbos_amd.ruletable.build import rule_table_from_policies  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
slice_req = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
thread_counts = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [1, 8, 32, 64, 128]
inner = int(sys.argv[6]) if len(sys.argv) > 6 else 1

pol, reqs = {"C2": (workloads.c2_policies, workloads.c2_requests), "C3": (workloads.c3_policies, workloads.c3_requests)}[name]
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
inputs = reqs(n_requests=n).to_inputs()
data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
capi.init(0)
table = capi.Table(lt.blob)
itab = ingest.IngestTable(lt.blob)
hip, ing = capi.load(), ingest.load()

slices = []
for a in range(0, n, slice_req):
    b = min(n, a + slice_req)
    d = np.ascontiguousarray(data[int(off[a]):int(off[b])])
    o = np.ascontiguousarray(off[a:b + 1] - off[a])
    slices.append((d, o, b - a))
max_tuples = max(sum(len(i["actions"]) for i in inputs[a:a + slice_req]) for a in range(0, n, slice_req))
params = capi.CParams(1_700_000_000_000_000_000, capi.F_WANT_DERIVED_ROLES, 0)


def worker(k, stop, counts, phase):
    res = capi.Result(max_tuples, slice_req, ("policy", "scope", "status", "edr"))
    done = 0
    t_flat = t_gpu = t_asm = 0.0
    i = k
    while not stop.is_set():
        d, o, cnt = slices[i % len(slices)]
        i += 1
        hb, ho = C.c_void_p(), C.c_void_p()
        t0 = time.perf_counter()
        if ing.cbi_flatten_pb_mt(itab.h, d.ctypes.data, o.ctypes.data, cnt, b"default", b"", 1, inner, C.byref(hb)) != 0:
            raise RuntimeError(ing.cbi_last_error())
        t1 = time.perf_counter()
        view = ing.cbi_batch_view(hb)
        if hip.cbh_check_batch(table.h, view, C.byref(params), C.byref(res.c)) != 0:
            raise RuntimeError(hip.cbh_last_error())
        t2 = time.perf_counter()
        if ing.cbi_assemble_pb_mt(itab.h, hb, C.byref(res.c), d.ctypes.data, o.ctypes.data, cnt, b"default", inner, C.byref(ho)) != 0:
            raise RuntimeError(ing.cbi_last_error())
        t3 = time.perf_counter()
        done += view.contents.n_tuples
        ing.cbi_outputs_free(ho)
        ing.cbi_batch_free(hb)
        t_flat += t1 - t0; t_gpu += t2 - t1; t_asm += t3 - t2
    counts[k] = done
    phase[k] = (t_flat, t_gpu, t_asm)


out = {"workload": name, "requests": n, "slice_requests": slice_req, "wire_bytes_per_request": float(data.size) / n, "runs": []}
for T in thread_counts:
    stop = threading.Event()
    counts, phase = [0] * T, [None] * T
    ths = [threading.Thread(target=worker, args=(k, stop, counts, phase)) for k in range(T)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    time.sleep(seconds)
    stop.set()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    tot = [sum(p[j] for p in phase) for j in range(3)]
    s = sum(tot) or 1.0
    run = {"threads": T, "inner_threads": inner, "decisions_per_s": sum(counts) / dt,
           "share_flatten": tot[0] / s, "share_gpu_roundtrip": tot[1] / s, "share_assemble": tot[2] / s}
    out["runs"].append(run)
    print(json.dumps(run), flush=True)
print(json.dumps(out))
