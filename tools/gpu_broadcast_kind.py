import sys; sys.path.insert(0, "/root/repo")
from cerbos_amd import capi, workloads
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c2_policies())))
for devs in ([0], [0, 0]):
    capi.init(devs)
    t = capi.Table(lt.blob)
    print("devices", devs, "-> broadcast kind:", t.broadcast_kind())
    t.close()
