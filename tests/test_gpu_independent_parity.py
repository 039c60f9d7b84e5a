"""GPU tier: full-size parity that does not pass through the product's lowering or flattener on the checker's side.

tests/test_gpu_synthetic.py compares every tuple of C2-C4 with oracle/ccheck.cpp, which walks the LOWERED image and the
FLATTENED batch: it proves kernel == C++ walker on the same image.  Here the checker is oracle/check.py - the restatement
of check.go pinned on the reference's golden fixtures - working from the policy dicts and the CheckInput dicts, on the
host cores in parallel: >= 100 000 requests per configuration, taken as evenly spaced blocks of the FULL-SIZE batch the
GPU decides (so that whatever only shows at size - directory growth, string-pool size, chunking of the upload - lies
between the sampled requests and the answer), compared per action (effect, policy key, scope) and per request (effective
derived roles, presence of evaluation errors).

Two more stores push the table past what the kernels keep in LDS, so that the off-paths run on hardware:
more than 4096 table strings (class tables read from memory, cbh_check_flat.h cls_in_lds) and more than 256 scopes
(32-bit scope-chain scratch, chain8)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from cerbos_amd import capi, workloads
from cerbos_amd.columnar import Vocab
from cerbos_amd.engine import HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

pytestmark = pytest.mark.gpu
NOW = 1_700_000_000_000_000_000
API = "api.cerbos.dev/v1"
_ORACLE = None   # set in the parent before the pool forks


def _pool_check(args):
    from oracle.check import EvalParams
    inputs, lenient = args
    p = EvalParams(now_ns=NOW, lenient_scope_search=lenient)
    out = []
    for inp in inputs:
        o = _ORACLE.check(inp, p)
        out.append(([(o["actions"][a]["effect"], o["actions"][a]["policy"], o["actions"][a].get("scope", "")) for a in inp["actions"]],
                    sorted(o.get("effectiveDerivedRoles") or []), bool(o.get("evaluationErrors"))))
    return out


class _Decoder(HipEvaluator):
    def __init__(self, lt):   # noqa: D401 - ids -> strings only, no device table of its own
        self.lt = lt


def _independent(docs, cr, n_sample, lenient=False, block=50, expect=None, make_table=None):
    global _ORACLE
    from oracle.check import RuleTableOracle
    rt = rule_table_from_policies(policies_from_docs(docs))
    lt = lower_rule_table(rt)
    if expect:
        expect(lt)
    table = make_table(lt) if make_table else capi.Table(lt.blob)
    batch = cr.to_batch(Flattener(lt))
    flags = capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0)
    got = table.check(batch, now_ns=NOW, flags=flags)   # input order
    db = table.upload(batch)
    plan = table.plan(db, flags)
    if db is not None:
        db.close()
    assert (got.status != capi.ST_UNSUPPORTED).all()
    n_blocks = max(1, min(n_sample // block, cr.n // block))
    starts = np.unique(np.linspace(0, cr.n - block, n_blocks).astype(np.int64))
    _ORACLE = RuleTableOracle(rt)
    chunks = [(cr.to_inputs(int(s), int(s) + block), lenient) for s in starts]
    nproc = max(1, min(64, (os.cpu_count() or 2) - 2))
    with mp.get_context("fork").Pool(nproc) as pool:
        want = pool.map(_pool_check, chunks, chunksize=4)
    dec = _Decoder(lt)
    eff_name = {capi.EFFECT_ALLOW: "EFFECT_ALLOW", capi.EFFECT_DENY: "EFFECT_DENY"}
    # (all configurations here carry 4 actions per request)
    checked = allowed = 0
    for s, (inputs, _), part in zip(starts, chunks, want):
        for j, (inp, (acts, edr, err)) in enumerate(zip(inputs, part)):
            r = int(s) + j
            assert len(inp["actions"]) == 4
            for k, (we, wp, ws) in enumerate(acts):
                t = 4 * r + k
                sc = int(got.scope[t])
                have = (eff_name[int(got.effect[t])], dec._policy_string(int(got.policy[t]), inp, "default"), "" if sc == capi.NONE else lt.scopes[sc])
                assert have == (we, wp, ws), (r, k, inp)
                allowed += we == "EFFECT_ALLOW"
            mask = int(got.edr[r])
            assert sorted(n for i, n in enumerate(lt.dr_names) if (mask >> i) & 1) == edr, (r, inp)
            assert bool((got.status[4 * r:4 * r + 4] == capi.ST_CEL_ERROR).any()) == err, (r, inp)
            checked += 1
    table.close()
    assert 0.02 < allowed / (4.0 * checked) < 0.98
    return checked, plan, lt


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,n_requests", [("C2", 250_000), ("C3", 1_000_000), ("C4", 500_000), ("T", 250_000), ("C5W", 250_000)])
def test_full_size_against_the_policy_level_oracle(name, n_requests):
    """T = north_star's target set (100 policies / 10k rules with CEL conditions, 1M tuples); C5W = C5's table with principals
    of five to eight roles (the walk's 8 x 8 shape)."""
    pol_fn, req_fn = {"C2": (workloads.c2_policies, workloads.c2_requests), "C3": (workloads.c3_policies, workloads.c3_requests),
                      "C4": (workloads.c4_policies, workloads.c4_requests), "T": (workloads.t_policies, workloads.t_requests),
                      "C5W": (workloads.c5_policies, workloads.c5w_requests)}[name]
    checked, plan, _ = _independent(pol_fn(), req_fn(n_requests), 100_000 if name != "C5W" else 50_000)
    assert checked >= (100_000 if name != "C5W" else 50_000)
    assert ("walk2_wide" in plan) if name == "C5W" else plan.startswith("cbh_check_flat_kernel")


@pytest.mark.timeout(900)
def test_more_than_4096_table_strings():
    """4 500 kinds: the class tables do not fit the LDS staging (cbh_check_flat.h cls_in_lds is off)."""
    def expect(lt):
        assert len(lt.strings) > 4096, len(lt.strings)
    docs = workloads.c4_policies(seed=11, n_policies=13_500, rules_per_policy=4)
    cr = workloads.c4_requests(200_000, seed=11, n_policies=13_500)
    checked, plan, _ = _independent(docs, cr, 20_000, expect=expect)
    assert checked >= 20_000 and plan.startswith("cbh_check_flat_kernel")


def _deep_scope_store(seed=12, n_kinds=24):
    """301 scopes (root, 12 x 6 x 3 tree); every kind has policies at ~50 of them (ancestors always included)."""
    rng = np.random.default_rng(seed)
    pool = workloads.c4_condition_pool()
    scopes = [""] + ["t%d" % i for i in range(12)] + ["t%d.u%d" % (i, j) for i in range(12) for j in range(6)] + \
             ["t%d.u%d.v%d" % (i, j, k) for i in range(12) for j in range(6) for k in range(3)]
    docs = []
    for kd in range(n_kinds):
        chosen = {""}
        for leaf in rng.choice(scopes[85:], size=22, replace=False):
            parts = str(leaf).split(".")
            chosen.update(".".join(parts[:d]) for d in range(1, 4))
        for scope in sorted(chosen):
            rules = []
            for _ in range(int(rng.integers(2, 7))):
                rule = {"actions": [str(a) for a in rng.choice(workloads.C4_ACTIONS, size=int(rng.integers(1, 4)), replace=False)],
                        "roles": [str(r) for r in rng.choice(workloads.C4_ROLES, size=int(rng.integers(1, 4)), replace=False)],
                        "effect": "EFFECT_DENY" if rng.random() < 0.15 else "EFFECT_ALLOW"}
                if rng.random() < 0.4:
                    rule["condition"] = pool[int(rng.integers(0, len(pool)))]
                rules.append(rule)
            rp = {"resource": "k%04d" % kd, "version": "default", "rules": rules}
            if scope:
                rp["scope"] = scope
                if rng.random() < 0.25:
                    rp["scopePermissions"] = "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS"
            docs.append({"apiVersion": API, "resourcePolicy": rp})
    return docs, scopes


@pytest.mark.timeout(900)
@pytest.mark.parametrize("lenient", [False, True])
def test_more_than_256_scopes(lenient):
    """301 scopes: scope-chain scratch in 32-bit entries (chain8 off), scope indices beyond a byte in every output."""
    def expect(lt):
        assert len(lt.scopes) > 256, len(lt.scopes)
    docs, scopes = _deep_scope_store()
    n = 200_000
    cr = workloads.c4_requests(n, seed=12, n_policies=72)   # 24 kinds + the unknown one
    rng = np.random.default_rng(120)
    req_scopes = scopes + ["t3.u2.v1.w9", "zz", "t11.u5.v2.x.y"]
    cr.resource_scope = Vocab(req_scopes, rng.integers(0, len(req_scopes), n))
    checked, plan, lt = _independent(docs, cr, 20_000, lenient=lenient, expect=expect)
    assert checked >= 20_000 and plan.startswith("cbh_check_flat_kernel")
