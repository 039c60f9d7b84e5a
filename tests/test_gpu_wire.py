"""GPU tier: the device flattener through the C ABI - cbh_wire_flatten -> cbh_check_resident -> cbh_result_download ->
cbh_wire_spans_download -> cbi_assemble_wire_pb - against the host road (cbi_flatten_pb -> cbh_check_batch -> cbi_assemble_pb)
on the same serialized CheckInputs: decisions tuple by tuple, then the serialized CheckOutputs byte by byte."""
import numpy as np
import pytest

from cerbos_amd import capi, wire, workloads
from cerbos_amd.ingest import IngestTable
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.lower.celc import LoweringError
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from test_fuzz_parity import _policies, _requests

pytestmark = pytest.mark.gpu
NOW = 1_700_000_000_000_000_000
FIELDS = ("effect", "policy", "scope", "status", "edr")


def _both_roads(lt, inputs, flags=0):
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    table, it = capi.Table(lt.blob), IngestTable(lt.blob)
    try:
        hb = it.flatten_pb(data, off)
        want_dev = table.check(hb, now_ns=NOW, flags=flags | capi.F_WANT_DERIVED_ROLES, device_order=True)
        want_out, want_flags = it.assemble_pb(hb, want_dev, data, off)
        want = want_dev.to_input_order(hb)
        db = table.wire_flatten(data, off)
        table.launch(db, now_ns=NOW, flags=flags | capi.F_WANT_DERIVED_ROLES)
        have = table.download(db)
        for f in FIELDS:
            a, b = getattr(want, f), getattr(have, f)
            assert np.array_equal(a, b), (f, int(np.flatnonzero(a != b)[0]) if a.shape == b.shape else (a.shape, b.shape))
        have_out, have_flags = it.assemble_wire_pb(have, data, off, table.wire_spans(db))
        assert have_out == want_out and np.array_equal(have_flags, want_flags)
        dev_out, dev_flags = table.wire_outputs(db, cap=64)     # the GPU writes the answers (a first buffer that is too small)
        assert dev_out == want_out and np.array_equal(dev_flags, want_flags)
        info = db.wire_info
        db.close()
        return info
    finally:
        table.close()
        it.close()


@pytest.mark.parametrize("name,n", [("C2", 20_000), ("C3", 20_000), ("C4", 6_000), ("C5", 20_000)])
def test_workloads_by_both_roads(name, n):
    pol = getattr(workloads, name.lower() + "_policies")
    reqs = getattr(workloads, name.lower() + "_requests")
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
    inputs = reqs(n_requests=n).to_inputs()
    rng = np.random.default_rng(1)
    inputs = [inputs[k] for k in rng.permutation(n)]   # arrival order: the routes thoroughly mixed
    info = _both_roads(lt, inputs)
    assert info["n_requests"] == n and info["n_host"] == 0
    assert (info["n_routes"] > 1) == (name != "C2")   # grouped by route on the device (C2 has one route: nothing to group)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_stores_by_both_roads(seed):
    rng = np.random.default_rng(10_000 + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    inputs = [i for i in _requests(rng, 1500) if len(i.get("actions") or []) <= 64]
    for flags in (0, capi.F_LENIENT_SCOPE_SEARCH):
        _both_roads(lt, inputs, flags)


def test_a_quarter_million_requests():
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c2_policies())))
    info = _both_roads(lt, workloads.c2_requests(n_requests=250_000).to_inputs())
    assert info["n_tuples"] == 1_000_000


def test_messages_for_the_host_flattener_and_malformed_ones():
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c2_policies())))
    inputs = workloads.c2_requests(n_requests=500).to_inputs()
    table = capi.Table(lt.blob)
    try:
        wide = list(inputs)
        wide[77] = dict(wide[77], actions=["a%d" % k for k in range(65)])
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in wide])
        with pytest.raises(capi.HostFlattenerNeeded):
            table.wire_flatten(data, off)
        msgs = [wire.encode_check_input(i) for i in inputs]
        msgs[300] = msgs[300][:-2]
        data, off = wire.pack_messages(msgs)
        with pytest.raises(capi.HipEngineError, match="index 300"):
            table.wire_flatten(data, off)
        db = table.wire_flatten(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
        assert db.n_tuples == 0
        db.close()
    finally:
        table.close()


def test_evaluator_bytes_path_takes_the_device_road():
    """HipEvaluator.check_pb: device road (default) == host road, with the trace pass's evaluation_errors / outputs appended"""
    from cerbos_amd.engine import HipEvaluator
    ev = HipEvaluator.from_policies(workloads.c5_policies())
    try:
        inputs = workloads.c5_requests(n_requests=3000).to_inputs()
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
        want, wflags = ev.check_pb(data, off, now_ns=NOW, trace=True, device_ingest=False)
        assert ev.last_road == "host"
        have, hflags = ev.check_pb(data, off, now_ns=NOW, trace=True)
        assert ev.last_road == "device"
        assert have == want and np.array_equal(np.asarray(wflags), np.asarray(hflags))
    finally:
        ev.close()


@pytest.mark.timeout(300)
def test_the_road_in_one_call_gives_the_three_calls_bytes(total=70_000):
    """cbh_wire_check_pb (the device road in one call, cut into slices that run side by side) against cbh_wire_flatten +
    cbh_check_resident + cbh_wire_outputs: the same bytes, offsets and flags - at a size that takes four slices, at sizes that
    take one, with an output buffer that is too small at first, and a message for the host flattener in the third slice."""
    from cerbos_amd import wire, workloads
    from cerbos_amd.lower.blob import lower_rule_table
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies
    for pol_fn, req_fn in ((workloads.c5_policies, workloads.c5_requests), (workloads.c3_policies, workloads.c3_requests)):
        lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol_fn())))
        table = capi.Table(lt.blob)
        inputs = req_fn(total).to_inputs()
        msgs = [wire.encode_check_input(i) for i in inputs]
        flags = capi.F_WANT_DERIVED_ROLES
        for n in (total, total * 2 // 7, 130, 1, 0):
            data, off = wire.pack_messages(msgs[:n])
            db = table.wire_flatten(data, off)
            table.launch(db, now_ns=NOW, flags=flags)
            want, wflags = table.wire_outputs(db)
            db.close()
            got, gflags = table.wire_check_pb(data, off, now_ns=NOW, flags=flags)
            assert got == want and list(gflags) == list(wflags), (n, len(got))
        # a buffer that is too small at first: the binding grows it from `need`
        part = total * 4 // 7
        data, off = wire.pack_messages(msgs[:part])
        small = (np.empty(1000, dtype=np.uint8), np.empty(part + 1, dtype=np.uint64), np.empty(part, dtype=np.uint8))
        got, _ = table.wire_check_pb(data, off, now_ns=NOW, flags=flags, out=small)
        db = table.wire_flatten(data, off)
        table.launch(db, now_ns=NOW, flags=flags)
        want, _ = table.wire_outputs(db)
        db.close()
        assert got == want
        # a CheckInput with more than 64 actions is the host flattener's: the whole call says so
        at = total * 5 // 7   # (in the third of four slices)
        big = dict(inputs[at], actions=["a%d" % i for i in range(70)])
        data, off = wire.pack_messages(msgs[:at] + [wire.encode_check_input(big)] + msgs[at + 1:])
        with pytest.raises(capi.HostFlattenerNeeded):
            table.wire_check_pb(data, off, now_ns=NOW, flags=flags)
        table.close()


@pytest.mark.timeout(300)
def test_two_calls_in_flight_from_one_thread(total=40_000):
    """cbh_wire_check_pb_submit / _collect: two calls of the device road in flight at once from ONE caller thread (the second's
    uploads under the first's downloads), collected in either order, give each the bytes the synchronous call gives; a ticket whose
    output buffer is too small reports 2 at collect; a message for the host flattener reports 1 at collect - nothing is lost
    between the worker's thread and the collecting one."""
    from cerbos_amd import wire, workloads
    from cerbos_amd.lower.blob import lower_rule_table
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c5_policies())))
    table = capi.Table(lt.blob)
    inputs = workloads.c5_requests(total).to_inputs()
    msgs = [wire.encode_check_input(i) for i in inputs]
    flags = capi.F_WANT_DERIVED_ROLES
    half = total // 2
    parts = [wire.pack_messages(msgs[:half]), wire.pack_messages(msgs[half:])]
    want = [table.wire_check_pb(d, o, now_ns=NOW, flags=flags) for d, o in parts]

    def bufs(n, cap):
        return (np.empty(cap, dtype=np.uint8), np.empty(n + 1, dtype=np.uint64), np.empty(max(n, 1), dtype=np.uint8))
    for order in ((0, 1), (1, 0)):
        tickets = [table.wire_check_pb_submit(d, o, bufs(len(o) - 1, 320 * (len(o) - 1) + 4096), now_ns=NOW, flags=flags) for d, o in parts]
        for i in order:
            got, gflags = table.wire_check_pb_collect(tickets[i])
            assert got == want[i][0] and list(gflags) == list(want[i][1]), (order, i)
    # three rounds back to back, the way a server's loop would: submit, submit, collect, submit, collect, collect
    t0 = table.wire_check_pb_submit(*parts[0], bufs(half, 320 * half + 4096), now_ns=NOW, flags=flags)
    t1 = table.wire_check_pb_submit(*parts[1], bufs(total - half, 320 * (total - half) + 4096), now_ns=NOW, flags=flags)
    assert table.wire_check_pb_collect(t0)[0] == want[0][0]
    t2 = table.wire_check_pb_submit(*parts[0], bufs(half, 320 * half + 4096), now_ns=NOW, flags=flags)
    assert table.wire_check_pb_collect(t1)[0] == want[1][0]
    assert table.wire_check_pb_collect(t2)[0] == want[0][0]
    # a ticket is collected through the table it was issued for: another table's name is refused and leaves the ticket as it was
    other = capi.Table(lt.blob)
    t = table.wire_check_pb_submit(*parts[1], bufs(total - half, 320 * (total - half) + 4096), now_ns=NOW, flags=flags)
    with pytest.raises(capi.HipEngineError, match="another table"):
        other.wire_check_pb_collect(t)
    assert table.wire_check_pb_collect(t)[0] == want[1][0]
    other.close()
    # too small an output buffer: 2 at collect (the size needed in the message)
    t = table.wire_check_pb_submit(*parts[0], bufs(half, 1000), now_ns=NOW, flags=flags)
    with pytest.raises(capi.HipEngineError, match="too small"):
        table.wire_check_pb_collect(t)
    # a CheckInput with more than 64 actions is the host flattener's: 1 at collect, with the worker's message
    bad = dict(inputs[3], actions=["a%d" % k for k in range(70)])
    d, o = wire.pack_messages(msgs[:100] + [wire.encode_check_input(bad)] + msgs[100:200])
    t = table.wire_check_pb_submit(d, o, bufs(201, 320 * 201 + 4096), now_ns=NOW, flags=flags)
    with pytest.raises(capi.HostFlattenerNeeded, match="host flattener"):
        table.wire_check_pb_collect(t)
    table.close()
