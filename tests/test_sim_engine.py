"""CPU tier: the library's HOST side on the simulator (tests/sim_engine.py) - the bodies of the GPU tier's last-sorting tests, at
sizes a fiber scheduler finishes, through the same C ABI: what round 4 wrote after its GPU minutes were spent (request road in
slices, the audit trail's entry points, the resident trail) has at least run end to end before it meets the hardware."""
import pytest

from sim_engine import sim_engine


@pytest.fixture()
def engine():
    with sim_engine() as capi:
        yield capi


def test_request_road_service_cases(engine):
    import test_zz_gpu_request_road as t
    t.test_service_level_cases_requests_in_outputs_out()
    t.test_a_malformed_request_is_named()


@pytest.mark.parametrize("name,n", [("C2", 700), ("C5", 500)])
def test_request_road_bytes(engine, name, n):
    import test_zz_gpu_request_road as t
    t.test_requests_give_the_bytes_their_check_inputs_give(name, n)


def test_request_road_audit_trail_golden_store(engine):
    import test_zz_gpu_request_road as t
    t.test_the_audit_trail_of_every_request_golden_store()


@pytest.mark.parametrize("name,n", [("C2", 600), ("C5", 400)])
def test_request_road_audit_trail_at_size(engine, name, n):
    import test_zz_gpu_request_road as t
    t.test_the_audit_trail_of_every_request_at_size(name, n)


def test_trail_decision_logs_and_oracle(engine):
    import test_zz_gpu_effective_policies as t
    t.test_the_reference_s_decision_logs_and_the_oracle_by_input()
    t.test_fuzz_stores_by_input(1)


def test_trail_c5_groups_and_resident_batch(engine):
    import test_zz_gpu_effective_policies as t
    t.test_c5_at_size_groups_and_decisions(n=1_400)
    t.test_a_resident_batch_keeps_its_trail_across_launches(n=900)


@pytest.mark.parametrize("name,n", [("c2", 500), ("c3", 500), ("t", 150)])
def test_trail_flat_tables(engine, name, n):
    import test_zz_gpu_effective_policies as t
    t.test_flat_tables_keep_their_trail_in_the_flat_kernels(name, n)


@pytest.mark.parametrize("name,n", [("c5", 400), ("c5w", 300)])
def test_trail_walk_tables(engine, name, n):
    import test_zz_gpu_effective_policies as t
    t.test_walk_tables_by_input(name, n)


def _in_own_process(body, env):
    """Switches the library reads once (static) need a process of their own."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path[:0] = [%r, %r]\nfrom sim_engine import sim_engine\nwith sim_engine() as capi:\n" % (here, os.path.dirname(here))
            + "".join("    " + line + "\n" for line in body.strip().splitlines()) + "print('sim ok')\n")
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "sim ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_the_roads_in_slices():
    """Four slices side by side (threads, chained uploads, bases from the slices' sizes): the input road's own GPU test at a
    fiftieth of its size, and the request road - with and without the trail - against the input road."""
    _in_own_process('''
import test_gpu_wire as w
w.test_the_road_in_one_call_gives_the_three_calls_bytes(total=1_400)
import test_zz_gpu_request_road as t
t.test_requests_give_the_bytes_their_check_inputs_give("C2", 1_500)
t.test_requests_give_the_bytes_their_check_inputs_give("C5", 1_200)
t.test_the_audit_trail_of_every_request_at_size("C5", 1_200)
t.test_the_audit_trail_of_every_request_golden_store()
''', {"CBH_WIRE_SLICE_MIN": "200", "CBH_WIRE_SLICE_MIN_BYTES": "20000"})


def test_the_pre_pass_in_two_kernels_through_the_library():
    """CBH_PRE_SPLIT=1 on the library's side (the site lists live with the batch): C5 decisions equal to the fused pre-pass's."""
    body = '''
import numpy as np
from cerbos_amd import workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
batch = Flattener(lt).flatten(workloads.c5_requests(n_requests=900).to_inputs(), "default", "")
table = capi.Table(lt.blob)
res = table.check(batch, now_ns=1_700_000_000_000_000_000, flags=capi.F_WANT_DERIVED_ROLES)
db = table.upload(batch)
for _ in range(2):
    table.launch(db, now_ns=1_700_000_000_000_000_000, flags=capi.F_WANT_DERIVED_ROLES)
r2 = table.download(db)
assert np.array_equal(res.effect, r2.effect) and np.array_equal(res.policy, r2.policy) and np.array_equal(res.status, r2.status)
np.save(OUT, np.concatenate([res.effect.astype(np.uint32), res.policy, res.status.astype(np.uint32)]))
'''
    import os
    import tempfile

    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        for mode in ("0", "1"):
            _in_own_process("OUT = %r\n" % os.path.join(d, "m%s.npy" % mode) + body, {"CBH_PRE_SPLIT": mode})
        assert np.array_equal(np.load(os.path.join(d, "m0.npy")), np.load(os.path.join(d, "m1.npy")))


def test_four_simulated_devices():
    """The library's bookkeeping per replica on a simulated node of four devices (CBH_SIM_DEVICES: they share the one memory):
    the image broadcast by peer copies, a large batch cut into one request range per device with every result at the caller's
    offsets (engine.go:332) - pageable and page-locked -, resident batches and the wire road on any replica.  (The RCCL branch
    needs real devices: tests/test_gpu_engine.py on a multi-GPU node.)"""
    _in_own_process('''
import numpy as np
from cerbos_amd import wire, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
NOW = 1_700_000_000_000_000_000
capi.init(0)
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c3_policies())))
cr = workloads.c3_requests(24_000)
batch = cr.to_batch(Flattener(lt))
flags = capi.F_WANT_DERIVED_ROLES
one = capi.Table(lt.blob)
want = one.check(batch, now_ns=NOW, flags=flags)
one.close()
capi.init([0, 1, 2, 3])
assert capi.num_devices() == 4
table = capi.Table(lt.blob)
assert table.broadcast_kind() == "peer-copy"
def same(a, b, what):
    for f in ("effect", "policy", "scope", "status", "edr"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), (what, f)
same(want, table.check(batch, now_ns=NOW, flags=flags), "sharded, pageable")
capi.pin_batch(batch)
same(want, table.check(batch, now_ns=NOW, flags=flags, pinned=True), "sharded, page-locked")
for dev in (1, 3):
    db = table.upload(batch, device_index=dev)
    table.launch(db, now_ns=NOW, flags=flags)
    same(want, table.download(db), "resident on replica %d" % dev)
    db.close()
inputs = cr.head(3_000).to_inputs()
data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
outs = []
for dev in (0, 2):
    db = table.wire_flatten(data, off, device_index=dev)
    table.launch(db, now_ns=NOW, flags=flags)
    outs.append(table.wire_outputs(db)[0])
    db.close()
assert outs[0] == outs[1] and len(outs[0]) == 3_000
table.close()
''', {"CBH_SIM_DEVICES": "4", "CBH_BCAST": "peer"})


def test_smoke_body(engine):
    """__graft_entry__.smoke() - what the driver runs first on the GPU box - against the simulator build."""
    import __graft_entry__ as g
    g.smoke()


def test_failing_allocations_and_copies_are_survived():
    """Fault injection on the simulator build: the k-th device / page-locked allocation (or asynchronous copy) from now on fails, for
    k = 0 .. - in the middle of a table load, a one-shot check, the sliced request road with its threads and chained uploads.  Every call then either succeeds or
    returns an error with a message; none hangs (a slice that gives up must release the ones waiting on it), and the next call with
    allocations allowed again answers as if nothing had happened."""
    _in_own_process('''
import ctypes as C
import numpy as np
from cerbos_amd import wire, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
NOW = 1_700_000_000_000_000_000
lib = capi.load()
lib.cbh_sim_set_alloc_budget.argtypes = [C.c_long]
lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
inputs = workloads.c5_requests(n_requests=240).to_inputs()
batch = Flattener(lt).flatten(inputs, "default", "")
groups = [inputs[k:k + 6] for k in range(0, len(inputs), 6)]
reqs = [wire.encode_check_resources_request({"requestId": "r", "principal": g[0]["principal"],
                                             "resources": [{"actions": i["actions"], "resource": i["resource"]} for i in g]}) for g in groups]
table = capi.Table(lt.blob)
flags = capi.F_WANT_DERIVED_ROLES
want = table.check(batch, now_ns=NOW, flags=flags)
want_road = table.wire_check_requests_pb(reqs, now_ns=NOW, flags=flags, trail=True)
lib.cbh_sim_set_copy_budget.argtypes = [C.c_long]
failed = survived = 0
for what, ks in (("load", range(0, 9)), ("check", range(0, 12, 3)), ("road", range(0, 56, 8)), ("road, copies", range(0, 40, 6))):
    for k in ks:
        fresh = None if what == "load" else capi.Table(lt.blob)     # (a fresh table: empty pools, every buffer a real allocation)
        if what == "road, copies":
            lib.cbh_sim_set_copy_budget(k)                           # the k-th asynchronous copy from now on fails
        else:
            lib.cbh_sim_set_alloc_budget(k)
        try:
            if what == "load":
                capi.Table(lt.blob).close()
            elif what == "check":
                got = fresh.check(batch, now_ns=NOW, flags=flags)
                assert np.array_equal(got.effect, want.effect)
            else:
                got = fresh.wire_check_requests_pb(reqs, now_ns=NOW, flags=flags, trail=True)
                assert got[0] == want_road[0] and np.array_equal(got[3], want_road[3])
            survived += 1
        except capi.HipEngineError as e:
            assert str(e), "an error without a message"
            failed += 1
        finally:
            lib.cbh_sim_set_alloc_budget(-1)
            lib.cbh_sim_set_copy_budget(-1)
        for t in (fresh, table):                                      # allocations allowed again: as if nothing had happened
            if t is not None:
                again = t.check(batch, now_ns=NOW, flags=flags)
                assert np.array_equal(again.effect, want.effect) and np.array_equal(again.policy, want.policy)
        if fresh is not None:
            if what.startswith("road"):
                road = fresh.wire_check_requests_pb(reqs, now_ns=NOW, flags=flags, trail=True)
                assert road[0] == want_road[0] and np.array_equal(road[3], want_road[3])
            fresh.close()
assert failed > 10 and survived >= 2, (failed, survived)
print("allocation failures survived: %d calls failed cleanly, %d had enough" % (failed, survived))
table.close()
''', {"CBH_WIRE_SLICE_MIN_BYTES": "20000"})


def test_four_waves_to_a_workgroup():
    """The workgroup shape of the GPU (tests/sim_engine.py build_four_waves): flat tables with and without the trail, C5 through the walk,
    its pre-pass and its trail form, the request road."""
    from sim_engine import build_four_waves
    _in_own_process('''
import test_zz_gpu_request_road as t
t.test_requests_give_the_bytes_their_check_inputs_give("C5", 900)
t.test_the_audit_trail_of_every_request_at_size("C2", 700)
import test_zz_gpu_effective_policies as e
e.test_walk_tables_by_input("c5w", 300)
e.test_flat_tables_keep_their_trail_in_the_flat_kernels("t", 200)
''', {"CBH_TEST_SIM_LIB": build_four_waves()})
