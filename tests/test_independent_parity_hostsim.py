"""CPU tier: the machinery of tests/test_gpu_independent_parity.py (block sampling, the policy-level oracle on a process
pool, decoding of policy / scope words) on the host simulation of the kernel source, at sizes that run in seconds - and
the two stores that push a table past the kernels' LDS staging (more than 4096 strings, more than 256 scopes)."""
import numpy as np

import hostsim_api
import test_gpu_independent_parity as T
from cerbos_amd import workloads
from cerbos_amd.columnar import Vocab


class _SimTable:
    def __init__(self, lt):
        self.lt = lt

    def check(self, batch, now_ns=0, flags=0):
        return hostsim_api.check(self.lt, batch, now_ns, flags)

    def upload(self, batch):
        return None

    def plan(self, db, flags):
        return "cbh_check_flat_kernel" if hostsim_api.last_kind() == 1 else "kind %d" % hostsim_api.last_kind()

    def close(self):
        pass


def test_more_than_256_scopes_on_the_simulator():
    docs, scopes = T._deep_scope_store()
    n = 1500
    cr = workloads.c4_requests(n, seed=12, n_policies=72)
    req_scopes = scopes + ["t3.u2.v1.w9", "zz", "t11.u5.v2.x.y"]
    cr.resource_scope = Vocab(req_scopes, np.random.default_rng(120).integers(0, len(req_scopes), n))
    for lenient in (False, True):
        checked, plan, lt = T._independent(docs, cr, n, lenient=lenient, make_table=_SimTable)
        assert checked == n and plan == "cbh_check_flat_kernel" and len(lt.scopes) > 256


def test_more_than_4096_strings_on_the_simulator():
    docs = workloads.c4_policies(seed=11, n_policies=13_500, rules_per_policy=4)
    cr = workloads.c4_requests(1500, seed=11, n_policies=13_500)
    checked, plan, lt = T._independent(docs, cr, 1500, make_table=_SimTable)
    assert checked == 1500 and plan == "cbh_check_flat_kernel" and len(lt.strings) > 4096
