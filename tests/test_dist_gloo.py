"""CPU tier, world_size 2 over gloo: the policy-image broadcast and the request sharding used by
bench.py --gpus N (the kernel runs through the host simulation here; RCCL on the GPU box)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import hostsim_api
    from cerbos_amd import dist as cdist
    from cerbos_amd import workloads
    from cerbos_amd.flatten import Flattener
    from cerbos_amd.lower.blob import lower_rule_table
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rt = rule_table_from_policies(policies_from_docs(workloads.c2_policies()))
    lt = lower_rule_table(rt)
    img = cdist.broadcast_image(lt.blob if rank == 0 else None, src=0)
    assert bytes(img.numpy().tobytes()) == lt.blob  # lowering is deterministic across ranks
    lt.blob = bytes(img.numpy().tobytes())           # evaluate with the *received* image
    cr = workloads.c2_requests(1000, seed=2)
    lo, hi = cdist.shard_range(cr.n, rank, world)
    inputs = cr.to_inputs(lo, hi)
    res = hostsim_api.check(lt, Flattener(lt).flatten(inputs), 1_700_000_000_000_000_000, 0)
    allow = torch.tensor([int((res.effect == 1).sum()), int(res.effect.size)], dtype=torch.int64)
    dist.all_reduce(allow)   # test-only aggregation (the product path has no such collective)
    if rank == 0:
        np.save(out, allow.numpy())
    dist.destroy_process_group()


def test_broadcast_and_shard(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim_api
    from cerbos_amd import workloads
    from cerbos_amd.flatten import Flattener
    from cerbos_amd.lower.blob import lower_rule_table
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies

    hostsim_api.build()
    out = str(tmp_path / "agg.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c2_policies())))
    cr = workloads.c2_requests(1000, seed=2)
    res = hostsim_api.check(lt, Flattener(lt).flatten(cr.to_inputs()), 1_700_000_000_000_000_000, 0)
    assert got.tolist() == [int((res.effect == 1).sum()), int(res.effect.size)]


def test_shard_range_covers_everything():
    from cerbos_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _worker_bytes(rank, world, port, out):
    """The widened path under sharding: each rank ingests its contiguous shard of serialized CheckInputs with
    the image it RECEIVED, decides, assembles serialized CheckOutputs; rank 0 collects them in input order."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pickle

    import torch.distributed as dist

    import hostsim_api
    from cerbos_amd import capi, wire, workloads
    from cerbos_amd import dist as cdist
    from cerbos_amd.ingest import IngestTable
    from cerbos_amd.lower.blob import lower_rule_table
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c3_policies())))
    img = cdist.broadcast_image(lt.blob if rank == 0 else None, src=0)
    lt.blob = bytes(img.numpy().tobytes())
    cr = workloads.c3_requests(900, seed=3)
    lo, hi = cdist.shard_range(cr.n, rank, world)
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in cr.to_inputs(lo, hi)])
    it = IngestTable(lt.blob)
    batch = it.flatten_pb(data, off)
    res = hostsim_api.check(lt, batch, 1_700_000_000_000_000_000, capi.F_WANT_DERIVED_ROLES, device_order=True)
    raw, flags = it.assemble_pb(batch, res, data, off)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((lo, raw, flags.tolist()), gathered, dst=0)   # test-only collection
    if rank == 0:
        with open(out, "wb") as fh:
            pickle.dump(sorted(gathered), fh)
    dist.destroy_process_group()


def test_sharded_bytes_path_equals_single_process(tmp_path):
    import pickle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim_api
    from cerbos_amd import capi, wire, workloads
    from cerbos_amd.ingest import IngestTable
    from cerbos_amd.lower.blob import lower_rule_table
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies

    hostsim_api.build()
    out = str(tmp_path / "outs.pkl")
    mp.spawn(_worker_bytes, args=(2, _free_port(), out), nprocs=2, join=True)
    with open(out, "rb") as fh:
        parts = pickle.load(fh)
    got = [r for _, raw, _ in parts for r in raw]
    got_flags = [f for _, _, fl in parts for f in fl]
    lt = lower_rule_table(rule_table_from_policies(policies_from_docs(workloads.c3_policies())))
    inputs = workloads.c3_requests(900, seed=3).to_inputs()
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    it = IngestTable(lt.blob)
    batch = it.flatten_pb(data, off)
    res = hostsim_api.check(lt, batch, 1_700_000_000_000_000_000, capi.F_WANT_DERIVED_ROLES, device_order=True)
    want, want_flags = it.assemble_pb(batch, res, data, off)
    assert [wire.decode_check_output(r) for r in got] == [wire.decode_check_output(r) for r in want]
    assert got_flags == want_flags.tolist()
    assert any(o["effectiveDerivedRoles"] for o in map(wire.decode_check_output, want))
