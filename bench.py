#!/usr/bin/env python3
"""Benchmark of the CheckResources hot path on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): 1 resource policy + 5 CEL
conditions, batches of 1M synthetic (principal, resource, action) tuples.  Every rank keeps a ROTATING SET of
such batches resident in HBM - 32 seeds, each resident four times in its own device buffers: 128 batches, 6.4 GB,
so neither the 32 MiB of L2 nor the 256 MiB Infinity Cache can hold what the next launch reads - and a "step" is
one sweep of the decision kernels over the whole set (one launch per 1M-tuple batch; 128 launches, ~2.2 ms: the
driver's 20 steps keep the GPU busy for ~50 ms).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Launches of different batches overlap on the device: resident batches are dealt to a few streams at upload
(cbh_table_set_resident_streams, default 4), so one launch's dispatch ramp fills what another's drain leaves idle.
`roofline.achieved` is therefore the rate the device sustains over the timed region; `roofline.kernel_ms` the average
begin-to-end time of one launch inside it; `roofline.serial` the same launches strictly one after the other on one stream.

Three rates, side by side in the line (the third: `wire_inclusive_decisions_per_s`, serialized CheckInputs in, serialized
CheckOutputs out by the device road, one host thread - what a server delivers): `value` = `resident_decisions_per_s`, the whole-job decision rate with the inputs
already resident in HBM when the timed region starts (what the kernel roofline is computed from), and
`pcie_inclusive_decisions_per_s`, SURVEY.md §8(d)'s service-level metric (1): the wall time of cbh_check_batch INCLUDING
the upload of the inputs and the download of the results (page-locked caller arrays, chunked three-stream pipeline), next
to the pageable-memory rate and the p50 of a 48-tuple round trip.  `--workload T` is north_star's named target set (100
policies / 10k rules with CEL conditions).

Multi-GPU: independent request shards per rank (weak scaling), no data-path collective; the only collective is the
one-time RCCL broadcast of the lowered policy image from rank 0.  `--inproc-gpus M` adds a single-process leg: one
engine over M devices, cbh_check_batch sharding each batch into contiguous request ranges behind the C ABI.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# 32 + 9*A/actions + 1 (C2: A=7, 4 actions)
ALG_BYTES_PER_DECISION = {"C1": 33.0, "C2": 49.0, "C3": 42.0, "C4": 47.0, "C5": 83.0, "T": 47.0, "C5W": 83.0}
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SERIAL_BATCHES = 16                     # resident batches of the one-stream leg (0.8 GB at C2: beyond the Infinity Cache)
# written by tools/gpu_final_r06.sh (this round's kernels); the older summaries stand in for workloads it does not cover
PMC_TRAFFIC_FILES = [os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_traffic.json", "r05_pmc_traffic_C3.json", "r04_pmc_traffic.json")]


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from separate rocprofv3 --pmc passes of THIS command (FETCH_SIZE and
    WRITE_SIZE do not fit one pass and a process cannot profile itself): read back from the committed summary,
    and the bench line says so in `traffic_source`.  None when no summary matches kernel and workload."""
    for path in PMC_TRAFFIC_FILES:
        try:
            with open(path) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        e = (d.get("workloads") or {}).get(workload)
        if not e or not all(k in kernel for k in (e.get("kernels") or [e.get("kernel", "?")])):
            continue
        return e["bytes_per_launch"], "profiles/%s: separate rocprofv3 --pmc passes of this command (%s)" % (os.path.basename(path), e.get("note", ""))
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("C1", "C2", "C3", "C4", "C5", "T", "C5W"), default="C2",
                    help="BASELINE.json config; the metric is quoted on C2 (the default), the others are side measurements; T = north_star's "
                         "target set: 100 policies / 10k rules with CEL conditions; C5W = C5 with principals of five to eight roles")
    ap.add_argument("--requests", type=int, default=None, help="requests per batch (default: the config's size: C1 10k x2, C2 250k x4, C3 1M x4 actions)")
    ap.add_argument("--batches", type=int, default=None, help="resident batches per GPU in the rotating set (default: enough for > 1 GB)")
    ap.add_argument("--replicas", type=int, default=None,
                    help="device copies of every seeded batch, each at its own HBM address and launched separately (default: 4 for C2, "
                         "so that a sweep is 128 launches over 6.4 GB and a 20-step run keeps the GPU busy for ~50 ms; 1 otherwise)")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="requests timed through the CPU oracle (~15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the PCIe-inclusive / latency legs (profiling runs)")
    ap.add_argument("--audit-trail", action="store_true",
                    help="measurement aid: every launch also keeps AuditTrail.EffectivePolicies (cbh_batch_set_trail, the trail forms of the "
                         "decision kernels: what a server with decision logs on pays); implies --no-side-legs except the one-stream leg")
    ap.add_argument("--serial-leg", action="store_true", help="measurement aid: keep the one-stream leg (the kernel by itself) under --no-side-legs")
    ap.add_argument("--check-first", action="store_true",
                    help="measurement aid (A/B runs): with --no-cpu-baseline, still compare every tuple of the first batch with oracle/ccheck.cpp after the timed region")
    ap.add_argument("--inproc-gpus", type=int, default=0, help="also time one engine over this many devices in THIS process")
    args = ap.parse_args()

    # stdout carries exactly one JSON line.  RCCL writes its banner and its warnings to file descriptor 1 from its own threads
    # (seen: a warning landing in the middle of the line), so everything else that writes there is sent to stderr and the
    # line goes out through a private copy of the original descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO"):
        os.environ["NCCL_DEBUG"] = "WARN"
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decision path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # The multi-GPU code path (RCCL init, image broadcast, adopt) is taken with ONE rank too, so that every run executes it and
    # `policy_bcast_ms` is a measured number (CBH_BENCH_NO_DIST=1: skip it at world 1).  A lone process brings its own rendezvous.
    use_dist = world > 1 or os.environ.get("CBH_BENCH_NO_DIST") != "1"
    dist_note = None
    if use_dist:
        if world == 1 and "MASTER_ADDR" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": str(local_rank)})
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        except Exception as e:   # a lone rank may go on without the collective; several ranks may not
            if world > 1:
                raise
            use_dist, dist_note = False, "RCCL init failed at world 1: %s" % str(e)[:200]

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if use_dist:
        dist.barrier()

    from cerbos_amd import capi, workloads
    from cerbos_amd.flatten import Flattener
    from cerbos_amd.lower.blob import lower_rule_table
    from cerbos_amd.policy.loader import policies_from_docs
    from cerbos_amd.ruletable.build import rule_table_from_policies

    capi.init(local_rank)
    wl = {"C1": (workloads.c1_policies, workloads.c1_requests, 10_000, 64, "RBAC-only template policy, no CEL"),
          "C2": (workloads.c2_policies, workloads.c2_requests, 250_000, 32, "1 resource policy + 5 CEL conditions"),
          "C3": (workloads.c3_policies, workloads.c3_requests, 1_000_000, 8,
                 "10 kinds x 20 rules, 4-level scope chain, 2 derived-role sets"),
          "C4": (workloads.c4_policies, workloads.c4_requests, 500_000, 12,
                 "1000 resource policies / 50k rules, 64-condition pool, Zipf kinds (one GPU's 2M of the 16M tuples)"),
          "C5": (workloads.c5_policies, workloads.c5_requests, 250_000, 16,
                 "C3 + principal overrides, action globs, role policies, nested map/list CEL (one GPU's 1M of 8M)"),
          "C5W": (workloads.c5_policies, workloads.c5w_requests, 250_000, 16,
                  "C5's table, principals with five to eight roles (cbh_walk2_wide_kernel; CBH_NO_WALK2_WIDE=1: the general walk) - a side line"),
          "T": (workloads.t_policies, workloads.t_requests, 250_000, 16,
                "north_star's target: 100 resource policies / 10k rules, 40 % with CEL conditions (64-condition pool), 1M tuples")}[args.workload]
    n_requests = args.requests or wl[2]
    n_seeded = max(1, args.batches or wl[3])
    replicas = max(1, args.replicas if args.replicas is not None else (4 if args.workload == "C2" and not args.batches else 1))
    n_batches = n_seeded * replicas   # launches per sweep
    rt = rule_table_from_policies(policies_from_docs(wl[0]()))
    lt = lower_rule_table(rt)   # deterministic: every rank derives the same host-side dictionaries

    # ---- policy image: lowered once, broadcast GPU->GPU over RCCL/xGMI
    bcast_ms = None
    if use_dist:
        from cerbos_amd import dist as cdist
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = cdist.broadcast_image(lt.blob if rank == 0 else None, src=0, device="cuda")
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
        table = capi.Table.adopt(img.data_ptr(), img.numel())   # `img` stays alive until exit
    else:
        table = capi.Table(lt.blob)

    # ---- this rank's rotating set (weak scaling: fixed tuples per GPU; every batch of every rank has its own seed)
    base_seed = {"C1": 1, "C2": 2, "C3": 3, "C4": 4, "C5": 5, "T": 7, "C5W": 5}[args.workload]
    fl = Flattener(lt)
    cr0 = batch0 = None
    dbatches, tuples_per_batch, resident_bytes = [], None, 0
    by_replica = [[] for _ in range(replicas)]
    serial_host = []   # host batches kept for the one-stream leg (rank 0)
    for k in range(n_seeded):
        cr = wl[1](n_requests, seed=base_seed + 1000 * k + rank)
        batch = cr.to_batch(fl)
        if k == 0:
            cr0, batch0 = cr, batch
            tuples_per_batch = batch.n_tuples
        assert batch.n_tuples == tuples_per_batch
        if rank == 0 and len(serial_host) < SERIAL_BATCHES:
            serial_host.append(batch)
        nbytes = sum(getattr(batch, f).nbytes for f in ("req_u32", "roles", "tuple_action", "col_tag", "col_val", "heap_tag",
                                                         "heap_val", "str_off", "str_bytes", "str_flags")) + 10 * batch.n_tuples + 8 * batch.n_requests
        for rep in range(replicas):   # every copy is its own set of device buffers: nothing of one launch serves another
            resident_bytes += nbytes
            by_replica[rep].append(table.upload(batch))
    dbatches = [db for rep in by_replica for db in rep]   # sweep order: all seeds of copy 0, then of copy 1, ...
    tuples = tuples_per_batch
    now = 1_700_000_000_000_000_000
    # the reference always computes effective derived roles (part of CheckOutput): so does every step here
    FLAGS = int(os.environ.get("CBH_BENCH_FLAGS", capi.F_WANT_DERIVED_ROLES))   # (override: experiments only)
    if args.audit_trail:
        FLAGS |= capi.F_WANT_EFFECTIVE_POLICIES
        for db in dbatches:
            table.set_trail(db, None, 1)   # one group per batch: one engine.Check call

    def sync_all():
        table.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def step():   # one sweep: cbh_check_resident for every batch of the set, in order (one crossing of the C ABI)
        table.launch_many(dbatches, now_ns=now, flags=FLAGS)

    for _ in range(args.warmup):
        step()
    sync_all()
    if args.warmup:
        table.kernel_time_ms()  # reset the kernel timer

    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    check_ms, resolve_ms = table.kernel_time_ms()

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    res = table.download(dbatches[0])
    eff = res.effect
    assert (res.status != capi.ST_UNSUPPORTED).all()
    streams = table.resident_streams

    # ---- the dominant kernel BY ITSELF: the same launches strictly one after the other (one stream), over their own device
    # copies of the first batches of the set - in the timed region above launches of different batches overlap on the
    # device, which lengthens each one's begin-to-end time while shortening the sweep
    serial = None
    if rank == 0 and streams > 1 and (not args.no_side_legs or args.serial_leg):
        table.set_resident_streams(1)
        sdb = [table.upload(hb) for hb in serial_host]
        if args.audit_trail:
            for db in sdb:
                table.set_trail(db, None, 1)
        for _ in range(2):
            table.launch_many(sdb, now_ns=now, flags=FLAGS)
        table.synchronize()
        table.kernel_time_ms()
        s0 = time.perf_counter()
        n_sweeps = max(3, min(args.steps, 400 // max(1, len(sdb))))
        for _ in range(n_sweeps):
            table.launch_many(sdb, now_ns=now, flags=FLAGS)
        table.synchronize()
        s_el = time.perf_counter() - s0
        s_ms, _ = table.kernel_time_ms()
        serial = {"streams": 1, "kernel_ms": s_ms, "launches": n_sweeps * len(sdb), "resident_batches": len(sdb),
                  "wall_ms_per_launch": s_el / (n_sweeps * len(sdb)) * 1e3}
        for db in sdb:
            db.close()
        table.set_resident_streams(streams)
    serial_host.clear()

    side = {}
    if args.audit_trail:
        args.no_side_legs = True   # (the one-shot legs have no trail to keep)
        if rank == 0:
            from cerbos_amd.engine import effective_policy_keys
            side["effective_policies_of_batch_0"] = len(effective_policy_keys(lt.policy_keys, table.trail(dbatches[0])[0]))
    if rank == 0 and not args.no_side_legs:
        # per-launch latency distribution (each launch synchronised; outside the timed region)
        lat = []
        for k in range(min(40, 4 * n_batches)):
            s0 = time.perf_counter()
            table.launch(dbatches[k % n_batches], now_ns=now, flags=FLAGS)
            table.synchronize()
            lat.append(time.perf_counter() - s0)
        side["p50_us_per_decision"] = float(np.median(lat)) / tuples * 1e6

        # p50 of a small synchronous round trip (SURVEY.md §8(d) metric 2): the first ~50 tuples as their own
        # one-shot batch (upload + kernels + download), the latency a single CheckResources call would see
        # (the dict flattener: a batch carries only the strings of its own requests, as a real call does)
        small = fl.flatten(cr0.to_inputs(0, max(1, min(n_requests, 50 // max(1, tuples // n_requests)))))
        import ctypes as C
        lib = capi.load()
        cb, prm = capi.make_cbatch(small, table.num_columns), capi.CParams(now, FLAGS, 0)
        sres = capi.Result(small.n_tuples, small.n_requests, ("policy", "scope", "status", "edr"))
        small_lat = []
        for _ in range(400):   # the C ABI call itself, as a cgo caller sees it (ctypes adds ~1 us)
            s0 = time.perf_counter()
            rc = lib.cbh_check_batch(table.h, C.byref(cb), C.byref(prm), C.byref(sres.c))
            small_lat.append(time.perf_counter() - s0)
            assert rc == 0
        side["p50_small_batch_roundtrip_us"] = float(np.median(small_lat[50:])) * 1e6
        side["p99_small_batch_roundtrip_us"] = float(np.percentile(small_lat[50:], 99)) * 1e6
        side["small_batch_tuples"] = int(small.n_tuples)
        py_lat = []
        for _ in range(100):   # the same through the Python wrapper (allocates result arrays, undoes the routing sort)
            s0 = time.perf_counter()
            table.check(small, now_ns=now, flags=FLAGS)
            py_lat.append(time.perf_counter() - s0)
        side["p50_small_batch_python_wrapper_us"] = float(np.median(py_lat[10:])) * 1e6

        # what the host link gives a plain page-locked copy (the PCIe-inclusive rate below is bounded by it)
        hbuf = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
        dbuf = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            dbuf.copy_(hbuf, non_blocking=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - c0)
        side["host_link_h2d_gbs"] = (64 << 20) / best / 1e9
        del hbuf, dbuf

        # SURVEY §8(d) metric (1): wall time of cbh_check_batch = upload + kernels + download of one 1M-tuple batch
        def oneshot(batch, want, pinned, reps):
            """The C ABI call itself, as a cgo caller makes it: the batch and result structs are built once, the clock is around
            cbh_check_batch (validation, upload, kernels, download)."""
            into = capi.Result(batch.n_tuples, batch.n_requests, want, pinned)
            cb1, prm1 = capi.make_cbatch(batch, table.num_columns), capi.CParams(now, FLAGS, 0)
            best = 1e9
            for _ in range(reps):      # the first call also grows the context's device block
                o0 = time.perf_counter()
                rc = lib.cbh_check_batch(table.h, C.byref(cb1), C.byref(prm1), C.byref(into.c))
                best = min(best, time.perf_counter() - o0)
                assert rc == 0, lib.cbh_last_error().decode()
            return batch.n_tuples / best
        full = ("policy", "scope", "status", "edr")
        side["pcie_inclusive_pageable_decisions_per_s"] = oneshot(batch0, (), False, 3)
        pb = capi.pin_batch(cr0.to_batch(fl))
        side["pcie_inclusive_decisions_per_s"] = oneshot(pb, (), True, 8)
        side["pcie_inclusive_all_outputs_decisions_per_s"] = oneshot(pb, full, True, 8)
        up_bytes = sum(getattr(pb, f).nbytes for f in ("roles", "tuple_action", "col_tag", "col_val", "heap_tag", "heap_val", "str_off",
                                                        "str_bytes", "str_flags")) + pb.req_u32.nbytes * (16 if lt.stats.get("reads_request_strings") else 10) // 16
        side["pcie_inclusive_upload_bytes"] = int(up_bytes)
        side["pcie_inclusive_frac_of_link"] = side["pcie_inclusive_decisions_per_s"] / (tuples / (up_bytes / (side["host_link_h2d_gbs"] * 1e9)))
        side["pcie_inclusive_note"] = ("cbh_check_batch wall time, one %d-tuple batch, page-locked arrays, effect-only results "
                                       "(all_outputs: + policy, scope, status, derived-role mask); pageable: ordinary numpy arrays; "
                                       "upload_bytes / host_link_h2d_gbs = the floor the link sets" % tuples)

        # Wire-inclusive (what a server delivers): serialized CheckInputs in, serialized CheckOutputs out, by the device road -
        # the raw bytes cross PCIe, the GPU flattens them (cbh_wire_flatten), decides (cbh_check_resident) and writes the
        # answers (cbh_wire_outputs); ONE host thread, page-locked buffers, slices of up to 128k requests of the first batch.
        try:
            from cerbos_amd import wire
            nw = n_requests      # the whole first batch: "decisions/sec at batch = 1M"
            w_inputs = cr0.to_inputs(0, nw)
            data, woff = wire.pack_messages([wire.encode_check_input(i) for i in w_inputs])
            pdata = capi.pinned_empty(data.size + 64, np.uint8)
            pdata[:data.size] = data
            out_cap = 320 * nw + 4096
            pout, poff, pfl = capi.pinned_empty(out_cap, np.uint8), capi.pinned_empty(nw + 1, np.uint64), capi.pinned_empty(nw + 1, np.uint8)
            wbest, wt = 1e9, 0
            for _ in range(8):     # cbh_wire_check_pb: the road in one call, the call's slices side by side (one caller thread)
                w0 = time.perf_counter()
                info, need = capi.CWireInfo(), C.c_size_t()
                rc = lib.cbh_wire_check_pb(table.h, 0, pdata.ctypes.data, woff.ctypes.data, nw, b"default", b"", None, 0, C.byref(prm),
                                           pout.ctypes.data, out_cap, poff.ctypes.data, pfl.ctypes.data, C.byref(need), C.byref(info))
                if rc != 0:
                    raise RuntimeError("cbh_wire_check_pb: rc %d: %s" % (rc, lib.cbh_last_error().decode()))
                wbest, wt = min(wbest, time.perf_counter() - w0), info.n_tuples
            n3 = min(nw, 131072)   # ... and the three calls in a row on one stream (round 3's figure), 131 072 messages per call
            t3best, t3 = 1e9, 0
            for _ in range(6):
                w0 = time.perf_counter()
                h, info, need = C.c_void_p(), capi.CWireInfo(), C.c_size_t()
                rc = lib.cbh_wire_flatten(table.h, 0, pdata.ctypes.data, woff.ctypes.data, n3, b"default", b"", None, 0, C.byref(h), C.byref(info))
                if rc != 0:
                    raise RuntimeError("cbh_wire_flatten: rc %d: %s" % (rc, lib.cbh_last_error().decode()))
                rc = lib.cbh_check_resident(table.h, h, C.byref(prm))
                rc = rc or lib.cbh_wire_outputs(table.h, h, pout.ctypes.data, out_cap, poff.ctypes.data, pfl.ctypes.data, C.byref(need))
                lib.cbh_batch_release(h)
                if rc != 0:
                    raise RuntimeError("device road: rc %d: %s" % (rc, lib.cbh_last_error().decode()))
                t3best, t3 = min(t3best, time.perf_counter() - w0), info.n_tuples
            side["wire_inclusive_three_calls_decisions_per_s"] = t3 / t3best
            # ... and the same ONE caller thread keeping TWO calls in flight (cbh_wire_check_pb_submit / _collect): the second call's
            # uploads run under the first's downloads.  Two sets of page-locked output buffers, the same input messages; ten calls,
            # the clock around all of them.
            try:
                sets = [(pout, poff, pfl), (capi.pinned_empty(out_cap, np.uint8), capi.pinned_empty(nw + 1, np.uint64), capi.pinned_empty(nw + 1, np.uint8))]
                def submit(k):
                    h = C.c_void_p()
                    rc_ = lib.cbh_wire_check_pb_submit(table.h, 0, pdata.ctypes.data, woff.ctypes.data, nw, b"default", b"", None, 0, C.byref(prm),
                                                       sets[k][0].ctypes.data, out_cap, sets[k][1].ctypes.data, sets[k][2].ctypes.data, C.byref(h))
                    if rc_ != 0:
                        raise RuntimeError("cbh_wire_check_pb_submit: rc %d: %s" % (rc_, lib.cbh_last_error().decode()))
                    return h
                def collect(h):
                    info_, need_ = capi.CWireInfo(), C.c_size_t()
                    rc_ = lib.cbh_wire_check_pb_collect(table.h, h, C.byref(need_), C.byref(info_))
                    if rc_ != 0:
                        raise RuntimeError("cbh_wire_check_pb_collect: rc %d: %s" % (rc_, lib.cbh_last_error().decode()))
                    return info_.n_tuples
                n_calls, best2 = 10, 1e9
                for _ in range(3):
                    w0 = time.perf_counter()
                    pend, done_t = [submit(0), submit(1)], 0
                    for c in range(2, n_calls):
                        done_t += collect(pend.pop(0))
                        pend.append(submit(c & 1))
                    while pend:
                        done_t += collect(pend.pop(0))
                    best2 = min(best2, (time.perf_counter() - w0) / n_calls)
                assert bytes(sets[1][0][:int(sets[1][1][nw])]) == bytes(sets[0][0][:int(sets[0][1][nw])]), "the two sets of output buffers differ"
                side["wire_inclusive_two_in_flight_decisions_per_s"] = (done_t / n_calls) / best2
            except Exception as e:
                side["wire_inclusive_two_in_flight_error"] = str(e)[:300]
            info, need = capi.CWireInfo(), C.c_size_t()
            rc = lib.cbh_wire_check_pb(table.h, 0, pdata.ctypes.data, woff.ctypes.data, nw, b"default", b"", None, 0, C.byref(prm),
                                       pout.ctypes.data, out_cap, poff.ctypes.data, pfl.ctypes.data, C.byref(need), C.byref(info))
            assert rc == 0
            # the answers the device wrote are the decisions of the resident run (first requests of the same batch)
            raw = pout[:int(poff[nw])].tobytes()
            k = 0
            for r in range(min(nw, 2000)):
                o = wire.decode_check_output(raw[int(poff[r]):int(poff[r + 1])])
                for a in w_inputs[r]["actions"]:
                    assert (o["actions"][a]["effect"] == "EFFECT_ALLOW") == (eff[k] == 1), "device-written CheckOutput differs from the resident decision"
                    k += 1
            side["wire_inclusive_decisions_per_s"] = wt / wbest
            side["wire_inclusive_note"] = ("serialized CheckInputs (%.0f B each) -> cbh_wire_check_pb (cbh_wire_flatten + cbh_check_resident + cbh_wire_outputs, "
                                           "the call's slices side by side) -> serialized CheckOutputs (%.0f B each), %d requests per call, ONE caller thread, "
                                           "best of 8; the host touches no message byte; three_calls: the three calls in a row, %d requests per call"
                                           % (data.size / nw, int(poff[nw]) / nw, nw, n3))
        except Exception as e:   # a side leg never takes the line down
            side["wire_inclusive_error"] = str(e)[:300]

    # north_star's own target configuration (100 policies / 10k rules with CEL conditions, 1M tuples per launch), timed in THIS
    # run whatever workload the line is quoted on: resident sweeps over 8 seeded batches, the kernel by itself on one stream, and
    # every tuple of the first batch against oracle/ccheck.cpp (the checker, after the timed region)
    target_t = None
    if rank == 0 and not args.no_side_legs and args.workload != "T":
        try:
            t_rt = rule_table_from_policies(policies_from_docs(workloads.t_policies()))
            t_lt = lower_rule_table(t_rt)
            t_table = capi.Table(t_lt.blob)
            t_fl = Flattener(t_lt)
            t_host = [workloads.t_requests(250_000, seed=7 + 1000 * k).to_batch(t_fl) for k in range(8)]
            t_db = [t_table.upload(b) for b in t_host]
            for _ in range(2):
                t_table.launch_many(t_db, now_ns=now, flags=FLAGS)
            t_table.synchronize()
            t_table.kernel_time_ms()
            t_steps = 12
            q0 = time.perf_counter()
            for _ in range(t_steps):
                t_table.launch_many(t_db, now_ns=now, flags=FLAGS)
            t_table.synchronize()
            t_el = time.perf_counter() - q0
            t_tuples = t_host[0].n_tuples
            t_rate = t_tuples * len(t_db) * t_steps / t_el
            t_res = t_table.download(t_db[0])
            t_kernel = t_table.plan(t_db[0], FLAGS)
            t_table.set_resident_streams(1)
            s_db = [t_table.upload(b) for b in t_host[:4]]
            for _ in range(2):
                t_table.launch_many(s_db, now_ns=now, flags=FLAGS)
            t_table.synchronize()
            t_table.kernel_time_ms()
            for _ in range(6):
                t_table.launch_many(s_db, now_ns=now, flags=FLAGS)
            t_table.synchronize()
            t_alone_ms, _ = t_table.kernel_time_ms()
            checked = None
            if not args.no_cpu_baseline:
                from oracle import ccheck
                want_t = ccheck.check(t_lt, t_host[0], now, FLAGS, threads=min(32, os.cpu_count() or 1))
                for name in ("effect", "policy", "scope"):
                    assert np.array_equal(getattr(t_res, name), getattr(want_t, name)), "target set: GPU %s differs from the C++ oracle" % name
                checked = "every tuple of the first batch (%d): effect, policy, scope identical to oracle/ccheck.cpp" % t_tuples
            alg_t = ALG_BYTES_PER_DECISION["T"]
            target_t = {"workload": "T: 100 resource policies / 10k rules, 40 %% with CEL conditions, %d tuples per launch" % t_tuples,
                        "resident_decisions_per_s": t_rate, "kernel": t_kernel, "streams": streams,
                        "sustained_frac": alg_t * t_rate / 1e9 / HBM_PEAK_GBS,
                        "by_itself_us": t_alone_ms * 1e3, "by_itself_frac": alg_t * t_tuples / (t_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "alg_bytes_per_decision": alg_t, "oracle_checked": checked, "north_star_target_decisions_per_s": 1e7}
            for db in t_db + s_db:
                db.close()
            t_table.close()
        except Exception as e:   # a side leg never takes the line down
            target_t = {"error": str(e)[:300]}

    if rank == 0 and args.inproc_gpus > 1:
        # one engine over several devices in this process: cbh_check_batch cuts the batch into request ranges
        ndev = min(args.inproc_gpus, torch.cuda.device_count())
        capi.init(list(range(ndev)))
        t2 = capi.Table(lt.blob)
        pb2 = capi.pin_batch(cr0.to_batch(fl))
        into = capi.Result(pb2.n_tuples, pb2.n_requests, (), True)
        best = 1e9
        for _ in range(6):
            o0 = time.perf_counter()
            t2.check(pb2, now_ns=now, flags=FLAGS, want=(), device_order=True, into=into)
            best = min(best, time.perf_counter() - o0)
        ok = bool(np.array_equal(into.to_input_order(pb2).effect, eff))
        side["inproc_multi_gpu"] = {"devices": ndev, "broadcast": t2.broadcast_kind(), "pcie_inclusive_decisions_per_s": pb2.n_tuples / best,
                                    "matches_single_gpu": ok}
        t2.close()
        capi.init(local_rank)

    first_checked = None
    if rank == 0 and args.no_cpu_baseline and args.check_first:
        from oracle import ccheck
        try:
            want_c = ccheck.check(lt, batch0, now, FLAGS, threads=min(32, os.cpu_count() or 1))
            covered = want_c.status != capi.ST_UNSUPPORTED
            for name in ("effect", "policy", "scope"):
                assert np.array_equal(getattr(res, name)[covered], getattr(want_c, name)[covered]), "GPU %s differs from the C++ oracle" % name
            first_checked = "%d of %d tuples of the first batch identical to oracle/ccheck.cpp" % (int(covered.sum()), covered.size)
        except ccheck.Unsupported:
            first_checked = "table outside oracle/ccheck.cpp"
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the oracle = checker + reported CPU baseline ("port": restatements, not the Go binary)
        #  * oracle/ccheck.cpp (scalar C++ restatement of check.go, -O2) over the WHOLE first batch: every
        #    output of every tuple must equal the GPU's; timed on 1 thread (repeated passes, ~10 s)
        #    and once on all host cores;
        #  * oracle/check.py (the restatement pinned on the reference's golden fixtures) over the first
        #    requests as the independent check of both.
        from oracle import ccheck
        from oracle.check import EvalParams, RuleTableOracle
        orc = RuleTableOracle(rt)
        sample = cr0.to_inputs(0, min(args.cpu_sample, n_requests, 5000))
        params = EvalParams(now_ns=now)
        p0 = time.perf_counter()
        outs = []
        for i in sample:   # bounded: the scan-based Python index is slow on large tables (C4)
            outs.append(orc.check(i, params))
            if len(outs) >= 20 and time.perf_counter() - p0 > 15.0:
                break
        sample = sample[:len(outs)]
        py_s = time.perf_counter() - p0
        want = np.array([1 if o["actions"][a]["effect"] == "EFFECT_ALLOW" else 2
                         for i, o in zip(sample, outs) for a in i["actions"]], dtype=np.uint8)
        assert np.array_equal(eff[:want.size], want), "GPU effects differ from the Python oracle on the sample"
        try:
            prep = ccheck.Prepared(lt, batch0)
            cres = prep.run(now_ns=now, flags=FLAGS, threads=1).to_input_order(batch0)
        except ccheck.Unsupported:   # a table outside the C++ restatement: the Python restatement is the baseline
            prep = None
            cpu = {"value": want.size / py_s, "unit": "decisions/s", "cores": 1, "kind": "port",
                   "sample": "first %d requests (%d tuples) of the same batch, Python restatement of check.go "
                             "(oracle/check.py), 1 thread, %.1f s"
                             % (len(sample), want.size, py_s)}
        if prep is not None:
            # requests whose decision path needs the general CEL interpreter are outside the C++ restatement
            # (it evaluates comparison trees only) and flagged by it: compared on the rest
            covered = cres.status != capi.ST_UNSUPPORTED
            for name in ("effect", "policy", "scope"):
                assert np.array_equal(getattr(res, name)[covered], getattr(cres, name)[covered]), "GPU %s differs from the C++ oracle" % name
            reps, c0 = 0, time.perf_counter()
            while reps < 3 or time.perf_counter() - c0 < 10.0:
                prep.run(now_ns=now, flags=FLAGS, threads=1, want=())
                reps += 1
            cpu_s = time.perf_counter() - c0
            ncpu = os.cpu_count() or 1
            m0 = time.perf_counter()
            prep.run(now_ns=now, flags=FLAGS, threads=ncpu, want=())
            mt_s = time.perf_counter() - m0
            cpu = {"value": tuples * reps / cpu_s, "unit": "decisions/s", "cores": 1, "kind": "port",
                   "sample": "%d passes over the first %d-tuple batch of the set, scalar C++ restatement of check.go "
                             "(oracle/ccheck.cpp, g++ -O2), 1 thread, %.1f s; all %d host threads: %.3g decisions/s%s"
                             % (reps, tuples, cpu_s, ncpu, tuples / mt_s,
                                "" if covered.all() else "; %.1f %% of the tuples need general CEL programs, which the C++ "
                                "restatement flags instead of evaluating (compared and timed on the rest)" % (100.0 * (1.0 - covered.mean())))}

    if rank == 0:
        total = tuples * n_batches * world * args.steps
        alg = ALG_BYTES_PER_DECISION[args.workload]
        # Launches of different batches overlap on the device (`streams` > 1), so the dominant kernel's rate is what the
        # device sustains over the timed region: algorithmic bytes of all launches of this GPU / the region's wall time.
        # `kernel_ms` is the average begin-to-end time of one launch INSIDE that region (HIP events on its own stream;
        # what rocprofv3's kernel trace of this command shows) - with overlap, longer than region / launches.
        # `serial` repeats the figure the earlier rounds reported: one stream, achieved = bytes / that launch's duration.
        achieved = alg * tuples * n_batches * args.steps / elapsed / 1e9 if streams > 1 else alg * tuples / (check_ms * 1e-3) / 1e9
        if serial is not None:
            serial["achieved"] = alg * tuples / (serial["kernel_ms"] * 1e-3) / 1e9
            serial["frac"] = serial["achieved"] / HBM_PEAK_GBS
        # the kernels cbh_check_resident launches for this table / batch / mode (cbh_plan_describe); kernel_ms spans the whole
        # plan - from the start of its first kernel to the end of its last
        max_act, max_roles = int(batch0.req_u32[9].max()), int(batch0.req_u32[7].max())
        kernel = table.plan(dbatches[0], FLAGS)
        if kernel == "cbh_check_kernel*":
            kernel = "cbh_check_kernel" + ("" if lt.stats["generic_programs"] else "_leaf") + \
                     (("_a4" if (max_act <= 4 and not lt.stats["generic_programs"]
                                 and lt.stats["kernel_features"]) else "_a32") + lt.stats["kernel_features"]
                      if max_act <= 32 else "")
        traffic, traffic_source = pmc_traffic(kernel, args.workload) if n_requests == wl[2] else (None, None)
        out = {
            "metric": "CheckResources decisions/sec at batch=1M; p50 per-decision us",
            "value": total / elapsed,
            "unit": "decisions/s",
            "value_is": "resident_decisions_per_s: whole-job rate with the inputs resident in HBM when the timed region starts (the "
                        "measurement contract of this tier: the PCIe-inclusive rate is never `value`); SURVEY.md 8(d) metric (1) - "
                        "cbh_check_batch including H2D / D2H at batch = 1M - is `pcie_inclusive_decisions_per_s` of this same line",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 ids + f64 attributes (CEL int64/uint64/double)",
            "data": "synthetic",
            "config": {"workload": "%s: %s; per GPU a rotating set of %d resident batches x %d tuples (%d requests) - %d seeded "
                                   "batches (seed per batch and rank), each resident %d time(s) in its own device buffers - one "
                                   "launch per batch, one step = one sweep of the set (%.2f GB resident: beyond L2 and the 256 MiB "
                                   "Infinity Cache)"
                                   % (args.workload, wl[4], n_batches, tuples, n_requests, n_seeded, replicas, resident_bytes / 1e9),
                       "batch_tuples": tuples, "batches_per_step": n_batches, "seeded_batches": n_seeded, "replicas": replicas,
                       "audit_trail": bool(args.audit_trail),
                       "parallelism": "independent request shards per GPU, policy image broadcast once"},
            "resident_decisions_per_s": total / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel, "kernel_ms": check_ms, "launch_tuples": tuples,
                         "alg_bytes_per_decision": alg, "streams": streams,
                         "achieved_is": "algorithmic bytes of all launches / wall time of the timed region (launches overlap on %d streams)" % streams
                                        if streams > 1 else "algorithmic bytes of one launch / kernel_ms",
                         "serial": serial},
            "cpu_baseline": cpu,
            "first_batch_checked": first_checked,
            "target_T": target_t,
            "resolve_kernel_ms": resolve_ms,
            "allow_fraction": float((eff == 1).mean()),
            "policy_bcast_ms": bcast_ms,
            "policy_bcast_note": dist_note or ("RCCL broadcast of the %d-byte image from rank 0 (world %d) + cbh_table_adopt_device_image" % (len(lt.blob), world) if use_dist else "skipped (CBH_BENCH_NO_DIST=1)"),
        }
        out.update(side)
        print(json.dumps(out), file=json_out, flush=True)
    for db in dbatches:
        db.close()
    table.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
