"""TableManager keeps serving the published table when an update does not compile - the reference's
internal/ruletable/ruletable_test.go:33-140 (TestRuleTableManager: a valid policy answers ALLOW; the policy is then updated to
name a derived role no import defines; the request must still be ALLOW), with the decisions from the kernel source on the host
simulator (the manager's device tables need a GPU: tests/test_gpu_engine.py runs swaps under load there)."""
import pytest

from cerbos_amd import capi, manager
from cerbos_amd.engine import Conf
from cerbos_amd.policy.compile import CompileError
from cerbos_amd.policy.loader import policies_from_docs

API = "api.cerbos.dev/v1"
ROCK = {"apiVersion": API, "resourcePolicy": {"resource": "rock", "version": "default", "rules": [
    {"actions": ["throw"], "roles": ["user"], "effect": "EFFECT_ALLOW"}]}}
ROCK_BAD = {"apiVersion": API, "resourcePolicy": {"resource": "rock", "version": "default", "importDerivedRoles": ["special_roles"], "rules": [
    {"actions": ["throw"], "derivedRoles": ["special_user"], "effect": "EFFECT_ALLOW"}]}}
INPUT = {"requestId": "1", "resource": {"kind": "rock", "id": "1"}, "principal": {"id": "sam", "roles": ["user"]}, "actions": ["throw"]}


class _SimTable:
    """capi.Table's face over the host simulator: load, retain, close."""

    def __init__(self, blob):
        self.blob, self.refs = blob, [1]

    @classmethod
    def borrow(cls, other):
        other.refs[0] += 1
        me = cls.__new__(cls)
        me.blob, me.refs = other.blob, other.refs
        return me

    def close(self):
        self.refs[0] -= 1


def _effect(mgr):
    from test_hostsim_golden import HostSimEvaluator
    with mgr.acquire() as lease:
        ev = HostSimEvaluator(lease.lowered, Conf())
        outs = ev.check([INPUT], now_ns=0)
        return outs[0]["actions"]["throw"]["effect"], lease.number


def test_valid_state_survives_an_update_that_does_not_compile(monkeypatch):
    monkeypatch.setattr(capi, "Table", _SimTable)
    mgr = manager.TableManager()
    assert mgr.swap_policies(policies_from_docs([ROCK])) == 1
    assert _effect(mgr) == ("EFFECT_ALLOW", 1)
    with pytest.raises(CompileError) as err:
        mgr.swap_policies(policies_from_docs([ROCK_BAD]))
    kinds = sorted(e.error for e in err.value.errors)
    assert kinds == ["import not found", "unknown derived role"]
    assert mgr.version == 1 and _effect(mgr) == ("EFFECT_ALLOW", 1)
    # the derived roles arrive: the update goes through and the answer follows it
    roles = {"apiVersion": API, "derivedRoles": {"name": "special_roles", "definitions": [{"name": "special_user", "parentRoles": ["admin"]}]}}
    assert mgr.swap_policies(policies_from_docs([ROCK_BAD, roles])) == 2
    assert _effect(mgr) == ("EFFECT_DENY", 2)
