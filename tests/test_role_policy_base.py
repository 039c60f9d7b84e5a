"""Index.Query's early exits in front of appendRolePolicyDenies (internal/ruletable/index/index.go:250-305).

The base bitmap = version & scope & resource & role & KIND_RESOURCE over ALL bindings (role-policy rows are
bindings too).  When it is empty at a scope the query returns before the synthetic role-policy DENYs are
built, so a role policy that exists at that scope but that nothing ties to the requested resource does NOT
deny there: the walk moves on to the parent scope.  No reference golden covers this layout (parity-unpinned:
derived by reading the reference), so the expectation below is written out by hand and the
oracle, the C++ restatement, the kernel source (host simulation) and the GPU kernel must all agree with it.
"""
import numpy as np
import pytest

from cerbos_amd import capi
from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from oracle.check import EvalParams, RuleTableOracle

NOW = 1_700_000_000_000_000_000


def _docs():
    rp = lambda res, actions, roles: {"apiVersion": "api.cerbos.dev/v1", "resourcePolicy": {   # noqa: E731
        "version": "default", "resource": res,
        "rules": [{"actions": actions, "effect": "EFFECT_ALLOW", "roles": roles}]}}
    return [
        rp("salary_record", ["view", "edit"], ["employee", "acme_admin"]),
        rp("leave_request", ["view", "approve"], ["employee", "acme_admin"]),
        # a resource policy AT scope acme for a third kind: gives the scope a resource-policy chain entry
        {"apiVersion": "api.cerbos.dev/v1", "resourcePolicy": {
            "version": "default", "resource": "leave_request", "scope": "acme",
            "scopePermissions": "SCOPE_PERMISSIONS_REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS",
            "rules": [{"actions": ["approve"], "effect": "EFFECT_ALLOW", "roles": ["acme_admin"]}]}},
        # the role policy covers ONLY leave_request
        {"apiVersion": "api.cerbos.dev/v1", "rolePolicy": {
            "role": "acme_admin", "scope": "acme", "parentRoles": ["employee"],
            "rules": [{"resource": "leave_request", "allowActions": ["view"]}]}},
    ]


def _inputs():
    mk = lambda kind, actions, roles, scope: {   # noqa: E731
        "requestId": "t", "actions": actions,
        "principal": {"id": "p1", "roles": roles, "scope": scope},
        "resource": {"kind": kind, "id": "r1", "scope": scope}}
    return [
        mk("salary_record", ["view", "edit", "delete"], ["acme_admin"], "acme"),
        mk("leave_request", ["view", "approve", "delete"], ["acme_admin"], "acme"),
        mk("salary_record", ["view"], ["employee"], "acme"),
        mk("salary_record", ["view"], ["acme_admin"], ""),
    ]


# hand-derived from index.go:214-336 + check.go:188-457
WANT = [
    # no binding at scope acme matches salary_record: base empty, no synthetic DENY; the root policy decides
    {"view": ("EFFECT_ALLOW", "resource.salary_record.vdefault", ""),
     "edit": ("EFFECT_ALLOW", "resource.salary_record.vdefault", ""),
     "delete": ("EFFECT_DENY", "resource.salary_record.vdefault", "")},
    # leave_request IS covered: view allowed by the role policy and the root policy, approve / delete are not
    # in the allow list -> synthetic DENY attributed to the role policy
    {"view": ("EFFECT_ALLOW", "resource.leave_request.vdefault/acme", ""),
     "approve": ("EFFECT_DENY", "role.acme_admin.vdefault/acme", "acme"),
     "delete": ("EFFECT_DENY", "role.acme_admin.vdefault/acme", "acme")},
    {"view": ("EFFECT_ALLOW", "resource.salary_record.vdefault", "")},
    {"view": ("EFFECT_ALLOW", "resource.salary_record.vdefault", "")},
]


def _norm(out):
    return {a: (e["effect"], e["policy"], e.get("scope", "")) for a, e in out["actions"].items()}


def _tables():
    rt = rule_table_from_policies(policies_from_docs(_docs()))
    return rt, lower_rule_table(rt)


def test_oracle_base_bitmap_early_exit():
    rt, _ = _tables()
    orc = RuleTableOracle(rt)
    for inp, want in zip(_inputs(), WANT):
        got = _norm(orc.check(inp, EvalParams(now_ns=NOW)))
        assert {a: v[0] for a, v in got.items()} == {a: v[0] for a, v in want.items()}, (inp, got)


def test_kernel_source_and_ccheck_agree_with_oracle():
    from hostsim_api import check as sim_check
    from oracle import ccheck
    from test_hostsim_golden import HostSimEvaluator
    rt, lt = _tables()
    orc = RuleTableOracle(rt)
    inputs = _inputs()
    ev = HostSimEvaluator(lt, Conf())
    outs = ev.check(inputs, now_ns=NOW)
    batch = Flattener(lt).flatten(inputs)
    cres = ccheck.check(lt, batch, NOW, capi.F_WANT_DERIVED_ROLES, 1)
    sres = sim_check(lt, batch, NOW, capi.F_WANT_DERIVED_ROLES)
    for name in ("effect", "policy", "scope"):
        assert np.array_equal(getattr(cres, name), getattr(sres, name)), name
    for inp, out in zip(inputs, outs):
        assert _norm(out) == _norm(orc.check(inp, EvalParams(now_ns=NOW))), inp


@pytest.mark.gpu
def test_gpu_base_bitmap_early_exit():
    rt, lt = _tables()
    orc = RuleTableOracle(rt)
    ev = HipEvaluator(lt, Conf())
    try:
        inputs = _inputs()
        for inp, out, want in zip(inputs, ev.check(inputs, now_ns=NOW), WANT):
            assert _norm(out) == _norm(orc.check(inp, EvalParams(now_ns=NOW))), inp
            assert {a: v[0] for a, v in _norm(out).items()} == {a: v[0] for a, v in want.items()}
    finally:
        ev.close()
