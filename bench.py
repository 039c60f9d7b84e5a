#!/usr/bin/env python3
"""Benchmark of the CheckResources hot path on MI355X.

A "step" is one pass of the decision kernels over one resident batch of synthetic tuples
(BASELINE.json configs[1]: 1 resource policy + 5 CEL conditions, 1M (principal, resource,
action) tuples per GPU).  Inputs are already in HBM when the timed region starts; the
PCIe-inclusive one-shot rate is reported separately (never as `value`).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: independent request shards per rank (weak scaling), no data-path collective; the
only collective is the one-time RCCL broadcast of the lowered policy image from rank 0.
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
ALG_BYTES_PER_DECISION = {"C1": 33.0, "C2": 49.0, "C3": 42.0, "C4": 47.0, "C5": 83.0}
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # written by tools/gpu_profile.sh


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE need separate passes, so bench.py cannot collect them on itself)."""
    try:
        with open(PMC_TRAFFIC) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None
    return d["bytes_per_launch"] if d.get("kernel") == kernel else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is ~40 us: the defaults keep the GPU busy long enough (~20 ms) for its clocks to settle
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", choices=("C1", "C2", "C3", "C4", "C5"), default="C2",
                    help="BASELINE.json config; the metric is quoted on C2 (the default), the others are side measurements")
    ap.add_argument("--requests", type=int, default=None, help="requests per GPU (default: the config's size: C1 10k x2, C2 250k x4, C3 1M x4 actions)")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="requests timed through the CPU oracle (~15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: keep RCCL's version banner (NCCL_DEBUG=VERSION/INFO) off it
    if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO"):
        os.environ["NCCL_DEBUG"] = "WARN"
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # CBH_BENCH_FORCE_DIST=1: take the multi-GPU code path (RCCL init, image broadcast, adopt) even with one rank
    use_dist = world > 1 or os.environ.get("CBH_BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decision path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

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
    wl = {"C1": (workloads.c1_policies, workloads.c1_requests, 10_000, "RBAC-only template policy, no CEL"),
          "C2": (workloads.c2_policies, workloads.c2_requests, 250_000, "1 resource policy + 5 CEL conditions"),
          "C3": (workloads.c3_policies, workloads.c3_requests, 1_000_000,
                 "10 kinds x 20 rules, 4-level scope chain, 2 derived-role sets"),
          "C4": (workloads.c4_policies, workloads.c4_requests, 500_000,
                 "1000 resource policies / 50k rules, 64-condition pool, Zipf kinds (one GPU's 2M of the 16M tuples)"),
          "C5": (workloads.c5_policies, workloads.c5_requests, 250_000,
                 "C3 + principal overrides, action globs, role policies, nested map/list CEL (one GPU's 1M of 8M)")}[args.workload]
    n_requests = args.requests or wl[2]
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

    # ---- this rank's shard (weak scaling: fixed tuples per GPU, different seed per rank)
    cr = wl[1](n_requests, seed={"C1": 1, "C2": 2, "C3": 3, "C4": 4, "C5": 5}[args.workload] + rank)
    batch = cr.to_batch(Flattener(lt))
    tuples = batch.n_tuples
    now = 1_700_000_000_000_000_000
    # the reference always computes effective derived roles (part of CheckOutput): so does every step here
    FLAGS = int(os.environ.get("CBH_BENCH_FLAGS", capi.F_WANT_DERIVED_ROLES))   # (override: experiments only)
    dbatch = table.upload(batch)

    def sync_all():
        table.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        table.launch(dbatch, now_ns=now, flags=FLAGS)
    sync_all()
    if args.warmup:
        table.kernel_time_ms()  # reset the kernel timer

    t0 = time.perf_counter()
    for _ in range(args.steps):
        table.launch(dbatch, now_ns=now, flags=FLAGS)
    sync_all()
    elapsed = time.perf_counter() - t0
    check_ms, resolve_ms = table.kernel_time_ms()

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # per-step latency distribution (each step synchronised; outside the timed region)
    lat = []
    for _ in range(min(args.steps, 20)):
        s0 = time.perf_counter()
        table.launch(dbatch, now_ns=now, flags=FLAGS)
        table.synchronize()
        lat.append(time.perf_counter() - s0)
    p50_us_per_decision = float(np.median(lat)) / tuples * 1e6

    # p50 of a small synchronous round trip (SURVEY.md §8(d) metric 2): the first ~50 tuples as their own
    # one-shot batch (upload + kernels + download), the latency a single CheckResources call would see
    small = cr.head(max(1, min(n_requests, 50 // max(1, tuples // n_requests)))).to_batch(Flattener(lt))
    small_lat = []
    for _ in range(60):
        s0 = time.perf_counter()
        table.check(small, now_ns=now, flags=FLAGS, want=())
        small_lat.append(time.perf_counter() - s0)
    p50_small_batch_us = float(np.median(small_lat[10:])) * 1e6

    res = table.download(dbatch)
    eff = res.effect
    assert (res.status != capi.ST_UNSUPPORTED).all()

    # PCIe-inclusive one-shot path (upload + kernels + download), for DESIGN.md; not `value`
    oneshot_s = 1e9
    for _ in range(3):      # the first call also grows the context's device block
        o0 = time.perf_counter()
        table.check(batch, now_ns=now, flags=FLAGS, want=())
        oneshot_s = min(oneshot_s, time.perf_counter() - o0)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the oracle = checker + reported CPU baseline ("port": restatements, not the Go binary)
        #  * oracle/ccheck.cpp (scalar C++ restatement of check.go, -O2) over the WHOLE batch: every
        #    output of every tuple must equal the GPU's; timed on 1 thread (repeated passes, ~10 s)
        #    and once on all host cores;
        #  * oracle/check.py (the restatement pinned on the reference's golden fixtures) over the first
        #    requests as the independent check of both.
        from oracle import ccheck
        from oracle.check import EvalParams, RuleTableOracle
        orc = RuleTableOracle(rt)
        sample = cr.to_inputs(0, min(args.cpu_sample, n_requests, 5000))
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
            prep = ccheck.Prepared(lt, batch)
            cres = prep.run(now_ns=now, flags=FLAGS, threads=1).to_input_order(batch)
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
                   "sample": "%d passes over the same %d-tuple batch, scalar C++ restatement of check.go "
                             "(oracle/ccheck.cpp, g++ -O2), 1 thread, %.1f s; all %d host threads: %.3g decisions/s%s"
                             % (reps, tuples, cpu_s, ncpu, tuples / mt_s,
                                "" if covered.all() else "; %.1f %% of the tuples need general CEL programs, which the C++ "
                                "restatement flags instead of evaluating (compared and timed on the rest)" % (100.0 * (1.0 - covered.mean())))}

    if rank == 0:
        total = tuples * world * args.steps
        alg = ALG_BYTES_PER_DECISION[args.workload]
        achieved = alg * tuples / (check_ms * 1e-3) / 1e9
        # the instantiation cbh_check_resident picks for this table / batch (cbh_engine.hip)
        kernel = "cbh_check_kernel" + ("" if lt.stats["generic_programs"] else "_leaf") + \
                 (("_a4" if (int(batch.req_u32[15].max()) <= 4 and not lt.stats["generic_programs"]
                             and lt.stats["kernel_features"]) else "_a32") + lt.stats["kernel_features"]
                  if int(batch.req_u32[15].max()) <= 32 else "")
        out = {
            "metric": "CheckResources decisions/sec at batch=1M; p50 per-decision us",
            "value": total / elapsed,
            "unit": "decisions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 ids + f64 attributes (CEL int64/uint64/double)",
            "data": "synthetic",
            "config": {"workload": "%s: %s, %d tuples/GPU (%d requests), seeded per rank"
                                   % (args.workload, wl[3], tuples, n_requests),
                       "parallelism": "independent request shards per GPU, policy image broadcast once"},
            "p50_us_per_decision": p50_us_per_decision,
            "p50_small_batch_roundtrip_us": p50_small_batch_us,
            "small_batch_tuples": int(small.n_tuples),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(kernel) if args.workload == "C2" and n_requests == wl[2] else None,
                         "kernel": kernel, "kernel_ms": check_ms,
                         "alg_bytes_per_decision": alg},
            "cpu_baseline": cpu,
            "resolve_kernel_ms": resolve_ms,
            "oneshot_pcie_inclusive_decisions_per_s": tuples / oneshot_s,
            "allow_fraction": float((eff == 1).mean()),
            "policy_bcast_ms": bcast_ms,
        }
        print(json.dumps(out))
    dbatch.close()
    table.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
