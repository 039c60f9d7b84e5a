"""One-off wider sweep of the wire kernels on the host simulator (tests/hostsim): for many fuzzed policy stores, the device
flattener against the host flattener value by value (arrival order, grouped by route), the decision kernels on both batches,
and the device assembler against the host assembler byte for byte.   python tools/wire_fuzz_sweep.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["CBH_WIRE_GROUP"] = "1"
import hostsim_api  # noqa: E402
import wire_device_util as wu  # noqa: E402
from cerbos_amd import wire  # noqa: E402
from cerbos_amd.ingest import IngestTable  # noqa: E402
from cerbos_amd.lower.blob import lower_rule_table  # noqa: E402
from cerbos_amd.lower.celc import LoweringError  # noqa: E402
from cerbos_amd.policy.loader import policies_from_docs  # noqa: E402
from cerbos_amd.ruletable.build import rule_table_from_policies  # noqa: E402
from test_fuzz_parity import _policies, _requests  # noqa: E402
from test_wire_device import _meta_flags  # noqa: E402

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 20_000), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
stores = requests = grouped = skipped = host_left = 0
NOW = 1_700_000_000_000_000_000
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    try:
        lt = lower_rule_table(rule_table_from_policies(policies_from_docs(_policies(rng))))
    except LoweringError:
        skipped += 1
        continue
    inputs = [i for i in _requests(rng, 260) if len(i.get("actions") or []) <= 64]
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    it = IngestTable(lt.blob)
    hb = it.flatten_pb(data, off, sort=False)
    rc, wb = wu.sim_flatten(lt, data, off)
    assert rc == 0 and wb.stats["first_bad"] == 0xFFFFFFFF, seed
    if wb.stats["n_host"]:
        host_left += 1
        it.close()
        continue
    wu.assert_same_requests(lt, hb, wb, bool(_meta_flags(lt) & wu.MF_READS_REQUEST_STRINGS))
    for flags in (4, 5):
        want = hostsim_api.check(lt, hb, now_ns=NOW, flags=flags, device_order=True)
        g = wb.grouped is not None
        have = hostsim_api.check(lt, wu.to_batch(lt, wb, grouped=g), now_ns=NOW, flags=flags, device_order=True)
        for f in ("effect", "policy", "scope", "status"):
            assert np.array_equal(getattr(want, f), getattr(have, f)), (seed, f)
        edr = have.edr[wb.grouped[3]] if g else have.edr
        assert np.array_equal(want.edr, edr), seed
        grouped += g
        act_off = np.concatenate([wb.req_u32[8], [wb.n_tuples]]).astype(np.uint32)
        a, af = it.assemble_wire_pb(want, data, off, (np.ascontiguousarray(wb.in_span), np.ascontiguousarray(wb.act_span), act_off))
        b, bf = wu.sim_outputs(lt, have, wb.n, edr_is_grouped=g)
        assert a == b and np.array_equal(af, bf), seed
    it.close()
    stores += 1
    requests += len(inputs)
print("wire sweep: %d stores (%d refused by the lowering, %d with messages for the host flattener), %d requests, %d grouped runs: clean"
      % (stores, skipped, host_left, requests, grouped))
