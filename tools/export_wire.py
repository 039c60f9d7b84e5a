"""Writes a workload as the files tools/e2e_bench.cpp reads: table.blob, messages.bin (serialized CheckInputs back to
back), offsets.bin (uint64[n + 1]).   python tools/export_wire.py C2 131072 /tmp/c2"""
import os
import sys

sys.path.insert(0, ".")
from cerbos_amd import wire, workloads  # noqa: E402
from cerbos_amd.lower.blob import lower_rule_table  # noqa: E402
from cerbos_amd.policy.loader import policies_from_docs  # noqa: E402
from cerbos_amd.ruletable.build import rule_table_from_policies  # noqa: E402

name, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
pol, reqs = {"C2": (workloads.c2_policies, workloads.c2_requests), "C3": (workloads.c3_policies, workloads.c3_requests),
             "C5": (workloads.c5_policies, workloads.c5_requests)}[name]
os.makedirs(out, exist_ok=True)
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
data, off = wire.pack_messages([wire.encode_check_input(i) for i in reqs(n_requests=n).to_inputs()])
open(os.path.join(out, "table.blob"), "wb").write(lt.blob)
data.tofile(os.path.join(out, "messages.bin"))
off.tofile(os.path.join(out, "offsets.bin"))
print("%d messages, %d bytes -> %s" % (n, data.size, out))
