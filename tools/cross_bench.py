"""Bulk evaluation as a cross product (cerbos_amd/cross.py) on the GPU: N principals x M resources x 4 actions from
N + M flattened messages - host time to build the batch, upload time, kernel time per launch.
    python tools/cross_bench.py [C2|C3] [N] [M] [launches]"""
import json
import sys
import time

sys.path.insert(0, ".")
from cerbos_amd import capi, workloads  # noqa: E402
from cerbos_amd.cross import cross_product_batch, effect_cube  # noqa: E402
from cerbos_amd.ingest import WireFlattener  # noqa: E402
from cerbos_amd.lower.blob import lower_rule_table  # noqa: E402
from cerbos_amd.policy.loader import policies_from_docs  # noqa: E402
from cerbos_amd.ruletable.build import rule_table_from_policies  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 50
pol, reqs = {"C2": (workloads.c2_policies, workloads.c2_requests), "C3": (workloads.c3_policies, workloads.c3_requests)}[name]
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
ins = reqs(n_requests=n + m).to_inputs()
principals, resources, actions = [i["principal"] for i in ins[:n]], [i["resource"] for i in ins[n:]], ins[0]["actions"]
capi.init(0)
table = capi.Table(lt.blob)
fl = WireFlattener(lt)
t0 = time.perf_counter()
batch = cross_product_batch(fl, lt.columns, principals, resources, actions)
t_build = time.perf_counter() - t0
t0 = time.perf_counter()
db = table.upload(batch)
t_upload = time.perf_counter() - t0
flags = capi.F_WANT_DERIVED_ROLES
for _ in range(5):
    table.launch(db, now_ns=1_700_000_000_000_000_000, flags=flags)
table.synchronize()
table.kernel_time_ms()
t0 = time.perf_counter()
for _ in range(launches):
    table.launch(db, now_ns=1_700_000_000_000_000_000, flags=flags)
table.synchronize()
t_loop = time.perf_counter() - t0
check_ms, _ = table.kernel_time_ms()
res = table.download(db)
cube = effect_cube(batch, res)
print(json.dumps({"workload": name, "principals": n, "resources": m, "actions": len(actions), "decisions": int(batch.n_tuples),
                  "messages_flattened": n + m, "host_build_s": t_build, "upload_s": t_upload, "kernel_ms": check_ms,
                  "decisions_per_s_resident": batch.n_tuples * launches / t_loop,
                  "decisions_per_s_including_build_and_upload": batch.n_tuples / (t_build + t_upload + check_ms * 1e-3),
                  "allow_fraction": float((cube == capi.EFFECT_ALLOW).mean())}))
