"""GPU tier: the walk's pre-pass in BOTH of its forms on the hardware.  The library reads CBH_PRE_SPLIT once per process (collector +
interpreter is the default since round 5, the fused pre-pass is CBH_PRE_SPLIT=0), so each form gets a process of its own; both decide
C5 and C5W (principal policies, role policies, globs, interpreter sites) as oracle/ccheck.cpp does wherever that restatement decides,
and as each other everywhere.  (The simulator tiers run both forms too: tests/test_pre_split.py, tests/test_sim_engine.py.)"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BODY = r'''
import os, sys
import numpy as np
from cerbos_amd import capi, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle import ccheck
NOW = 1_700_000_000_000_000_000
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
table = capi.Table(lt.blob)
out = []
for req_fn, flags in ((workloads.c5_requests, capi.F_WANT_DERIVED_ROLES), (workloads.c5w_requests, capi.F_WANT_DERIVED_ROLES | capi.F_LENIENT_SCOPE_SEARCH)):
    batch = req_fn(40_000).to_batch(Flattener(lt))
    got = table.check(batch, now_ns=NOW, flags=flags)
    db = table.upload(batch)
    plan = table.plan(db, flags)
    db.close()
    want_split = os.environ["CBH_PRE_SPLIT"] != "0"
    assert ("collect" in plan) == want_split, plan
    want = ccheck.check(lt, batch, NOW, flags, threads=4)
    assert (got.status != capi.ST_UNSUPPORTED).all()
    ok = want.status != capi.ST_UNSUPPORTED
    assert ok.mean() > 0.6
    for f in ("effect", "policy", "scope"):
        assert np.array_equal(getattr(got, f)[ok], getattr(want, f)[ok]), f
    out += [got.effect.astype(np.uint32), got.policy, got.scope, got.status.astype(np.uint32), got.edr.astype(np.uint64).view(np.uint32)]
table.close()
np.save(sys.argv[1], np.concatenate(out))
print("pre-pass ok")
'''


def test_c5_by_both_forms_of_the_pre_pass_on_the_device():
    if os.environ.get("CBH_TEST_SIM_ENGINE"):
        pytest.skip("the simulator build's own test of both forms is tests/test_sim_engine.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        for mode in ("0", "1"):
            env = dict(os.environ, CBH_PRE_SPLIT=mode, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
            r = subprocess.run([sys.executable, "-c", BODY, os.path.join(d, "m%s.npy" % mode)], env=env, capture_output=True, text=True, timeout=900, cwd=root)
            assert r.returncode == 0 and "pre-pass ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        a, b = np.load(os.path.join(d, "m0.npy")), np.load(os.path.join(d, "m1.npy"))
        assert np.array_equal(a, b), int(np.flatnonzero(a != b)[0])
