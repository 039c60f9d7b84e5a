"""GPU tier: the host side of the C ABI (cbh_engine.hip) - chunked / pinned one-shot pipeline, request ranges
sharded over several devices of one engine (exercised on a single GPU by naming it twice: two replicas, image
broadcast by peer copy), reference-counted tables and the versioned swap, input validation."""
import os
import threading

import numpy as np
import pytest

from cerbos_amd import capi, workloads
from cerbos_amd.flatten import Flattener
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

pytestmark = pytest.mark.gpu
NOW = 1_700_000_000_000_000_000
FIELDS = ("effect", "policy", "scope", "status", "edr")


def _lowered(pol_fn):
    return lower_rule_table(rule_table_from_policies(policies_from_docs(pol_fn())))


def _same(a, b, what):
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), (what, f)


@pytest.fixture
def two_replicas():
    capi.init([0, 0])
    yield
    capi.init(0)


@pytest.mark.parametrize("name,pol_fn,req_fn,n", [("C3", workloads.c3_policies, workloads.c3_requests, 150_000),
                                                 ("C5", workloads.c5_policies, workloads.c5_requests, 60_000)])
def test_pinned_chunked_pipeline_matches_other_paths(name, pol_fn, req_fn, n, monkeypatch):
    """One large batch decided four ways must give the same arrays: resident (upload / launch / download), one-shot
    from pageable memory (whole arrays, one launch), one-shot from page-locked memory in 64k-request chunks and in
    deliberately small, odd-sized chunks over the three streams."""
    lt = _lowered(pol_fn)
    table = capi.Table(lt.blob)
    batch = req_fn(n).to_batch(Flattener(lt))
    flags = capi.F_WANT_DERIVED_ROLES
    db = table.upload(batch)
    table.launch(db, now_ns=NOW, flags=flags)
    resident = table.download(db)
    db.close()
    pageable = table.check(batch, now_ns=NOW, flags=flags)
    _same(resident, pageable, "pageable one-shot")
    capi.pin_batch(batch)
    pinned = table.check(batch, now_ns=NOW, flags=flags, pinned=True)
    _same(resident, pinned, "pinned, default chunks")
    monkeypatch.setenv("CBH_CHUNK_REQUESTS", "4100")   # rounded up to 4160: chunks of 65 waves, ragged tail
    small = table.check(batch, now_ns=NOW, flags=flags, pinned=True)
    _same(resident, small, "pinned, small chunks")
    only_effect = table.check(batch, now_ns=NOW, flags=flags, want=(), pinned=True)
    assert np.array_equal(only_effect.effect, resident.effect)
    table.close()


def test_request_ranges_over_two_replicas(two_replicas):
    """The engine initialised with a device list: the image is broadcast (peer copy - RCCL wants distinct GPUs), a
    large batch is cut into one contiguous request range per replica and every result lands at the caller's
    offsets (engine.go:332); a batch in non-ascending tuple order is not split but still decided."""
    assert capi.num_devices() == 2
    lt = _lowered(workloads.c3_policies)
    table = capi.Table(lt.blob)
    assert table.broadcast_kind() == "peer-copy"
    fl = Flattener(lt)
    cr = workloads.c3_requests(100_000)
    batch = cr.to_batch(fl)
    flags = capi.F_WANT_DERIVED_ROLES
    want = None
    for dev in (0, 1):   # resident on either replica
        db = table.upload(batch, device_index=dev)
        table.launch(db, now_ns=NOW, flags=flags)
        table.synchronize()
        res = table.download(db)
        db.close()
        if want is None:
            want = res
        _same(want, res, "resident on replica %d" % dev)
    _same(want, table.check(batch, now_ns=NOW, flags=flags), "sharded, pageable")
    capi.pin_batch(batch)
    _same(want, table.check(batch, now_ns=NOW, flags=flags, pinned=True), "sharded, pinned")
    # small batches stay on the first replica
    small = cr.head(50).to_batch(fl)
    one = table.check(small, now_ns=NOW, flags=flags)
    assert one.effect.size == small.n_tuples
    # tuples in descending request order: still valid input (each request owns a contiguous slice)
    rev = cr.head(40_000).to_batch(fl)
    base = table.check(rev, now_ns=NOW, flags=flags, device_order=True)
    nr, act_cnt = rev.n_requests, rev.req_u32[9].copy()
    new_off = (np.cumsum(act_cnt[::-1])[::-1] - act_cnt).astype(np.uint32)   # request r's slice now sits after r+1's
    ta = np.empty_like(rev.tuple_action)
    for r in range(nr):
        ta[new_off[r]:new_off[r] + act_cnt[r]] = rev.tuple_action[rev.req_u32[8][r]:rev.req_u32[8][r] + act_cnt[r]]
    old_off = rev.req_u32[8].copy()
    rev.tuple_action = ta
    rev.req_u32[8] = new_off
    got = table.check(rev, now_ns=NOW, flags=flags, device_order=True)
    for r in range(0, nr, 997):
        assert np.array_equal(got.effect[new_off[r]:new_off[r] + act_cnt[r]], base.effect[old_off[r]:old_off[r] + act_cnt[r]])
    table.close()


def test_released_table_drains_inflight_batches():
    """cbh_table_release while other threads are inside cbh_check_batch on that table: each call holds a reference,
    the image is freed by whichever reference goes last (manager.go:86-124 swaps without waiting for readers)."""
    lt = _lowered(workloads.c2_policies)
    table = capi.Table(lt.blob)
    batch = workloads.c2_requests(120_000).to_batch(Flattener(lt))
    want = table.check(batch, now_ns=NOW).effect.copy()
    n_threads, rounds = 6, 6
    for _ in range(n_threads):
        table.retain()
    h = table.h.value
    lib = capi.load()
    errors = []
    started = threading.Barrier(n_threads + 1)

    def worker():
        try:
            started.wait()
            for _ in range(rounds):
                res = capi.Result(batch.n_tuples, batch.n_requests, ())
                cb = capi.make_cbatch(batch, table.num_columns)
                p = capi.CParams(NOW, 0, 0)
                rc = lib.cbh_check_batch(h, capi.C.byref(cb), capi.C.byref(p), capi.C.byref(res.c))
                if rc != 0 or not np.array_equal(res.to_input_order(batch).effect, want):
                    errors.append(rc)
        finally:
            lib.cbh_table_release(h)

    th = [threading.Thread(target=worker) for _ in range(n_threads)]
    for x in th:
        x.start()
    started.wait()
    table.close()   # the owner's reference goes while the workers are busy
    for x in th:
        x.join()
    assert not errors


def test_versioned_swap_under_load():
    """TableManager (cerbos_amd/manager.py = ruletable.Manager): requests run while the table is swapped between two
    policy sets; every response must be the answer of exactly the version it reports."""
    from cerbos_amd.manager import TableManager
    from oracle.check import EvalParams, RuleTableOracle
    docs_a = workloads.c2_policies()
    docs_b = workloads.c2_policies()
    for r in docs_b[0]["resourcePolicy"]["rules"]:   # version B flips every effect
        r["effect"] = "EFFECT_DENY" if r["effect"] == "EFFECT_ALLOW" else "EFFECT_ALLOW"
    rts = [rule_table_from_policies(policies_from_docs(d)) for d in (docs_a, docs_b)]
    lts = [lower_rule_table(rt) for rt in rts]
    inputs = workloads.c2_requests(3000).to_inputs()
    want = []
    for rt in rts:
        orc = RuleTableOracle(rt)
        want.append(np.array([1 if orc.check(i, EvalParams(now_ns=NOW))["actions"][a]["effect"] == "EFFECT_ALLOW" else 2
                              for i in inputs for a in i["actions"]], dtype=np.uint8))
    assert not np.array_equal(want[0], want[1])
    mgr = TableManager(lts[0])
    stop, errors, seen = threading.Event(), [], set()

    done = [0]

    def worker():
        while not stop.is_set():
            ver, _, _, res = mgr.check_batch(inputs, now_ns=NOW)
            seen.add(ver)
            done[0] += 1
            if not np.array_equal(res.effect, want[(ver - 1) % 2]):
                errors.append(ver)

    th = [threading.Thread(target=worker) for _ in range(4)]
    for x in th:
        x.start()
    from cerbos_amd.ruletable.proto import encode_rule_table
    wires = [encode_rule_table(rt) for rt in rts]   # the reference's artefact: serialized runtimev1.RuleTable
    import time
    for k in range(1, 9):
        if k % 3 == 0:
            mgr.swap(lts[k % 2])
        else:
            mgr.swap_pb(wires[k % 2])
        mark, t0 = done[0], time.time()          # (some answers under every version, however slow the machine)
        while done[0] < mark + 2 and time.time() - t0 < 120:
            time.sleep(0.001)
    stop.set()
    for x in th:
        x.join()
    mgr.close()
    assert not errors and len(seen) >= 2, (errors[:3], seen)


def test_invalid_batches_are_refused_not_dereferenced():
    lt = _lowered(workloads.c2_policies)
    table = capi.Table(lt.blob)
    fl = Flattener(lt)
    good = workloads.c2_requests(300).to_batch(fl)
    lib = capi.load()

    def call(batch, mutate):
        cb = capi.make_cbatch(batch, table.num_columns)
        mutate(cb, batch)
        res = capi.Result(batch.n_tuples, batch.n_requests, ())
        p = capi.CParams(NOW, 0, 0)
        rc = lib.cbh_check_batch(table.h, capi.C.byref(cb), capi.C.byref(p), capi.C.byref(res.c))
        h = capi.C.c_void_p()
        rc2 = lib.cbh_batch_upload(table.h, capi.C.byref(cb), capi.C.byref(h))
        if rc2 == 0:
            lib.cbh_batch_release(h)
        return rc, rc2

    assert call(good, lambda cb, b: None) == (0, 0)
    assert call(good, lambda cb, b: setattr(cb, "req_u32", None)) == (-1, -1)
    assert call(good, lambda cb, b: setattr(cb, "tuple_action", None)) == (-1, -1)
    assert call(good, lambda cb, b: setattr(cb, "n_columns", cb.n_columns + 1)) == (-1, -1)
    for field, bump in ((6, 10_000_000), (7, 10_000), (8, 4_000_000_000), (9, 65)):   # ROLE_OFF, ROLE_CNT, ACT_OFF, ACT_CNT
        bad = workloads.c2_requests(300).to_batch(fl)
        bad.req_u32[field][17] = bump
        assert call(bad, lambda cb, b: None) == (-1, -1), field
    assert b"outside the batch" in lib.cbh_last_error() or b"CBH_MAX_ACTIONS" in lib.cbh_last_error()
    table.close()


def _visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs two distinct GPUs: the in-library ncclBroadcast runs one rank per device")
def test_image_broadcast_over_rccl_and_the_wire_road_on_the_second_device():
    """With >= 2 distinct devices the table image reaches the others by ONE RCCL broadcast from C++ (cbh_engine.hip
    broadcast_image: ncclCommInitAll + ncclBroadcast over xGMI), and a replica that got its image that way decides - and
    flattens serialized CheckInputs - exactly as the first."""
    from cerbos_amd import wire
    capi.init([0, 1])
    try:
        lt = _lowered(workloads.c3_policies)
        table = capi.Table(lt.blob)
        assert table.broadcast_kind() == "rccl"
        inputs = workloads.c3_requests(20_000).to_inputs()
        data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
        res = []
        for dev in (0, 1):
            db = table.wire_flatten(data, off, device_index=dev)
            table.launch(db, now_ns=NOW, flags=capi.F_WANT_DERIVED_ROLES)
            res.append((table.download(db), table.wire_outputs(db)))
            db.close()
        _same(res[0][0], res[1][0], "wire road on device 1")
        assert res[0][1][0] == res[1][1][0]
        table.close()
    finally:
        capi.init(0)
