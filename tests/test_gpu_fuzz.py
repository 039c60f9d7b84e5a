"""GPU tier: the differential fuzz of tests/test_fuzz_parity.py through the real library - every
kernel feature class (plain / derived roles / globs / everything; leaf and interpreter; 4, 32 and 64
action masks) on hardware, against oracle/check.py: 24 seeds with the generator the tier started with and 24 with
the wider CEL surface of CONDITIONS_WIDE (arithmetic, ternaries, string functions, comprehensions, timestamps,
IP ranges, JWT claims)."""
import os

import numpy as np
import pytest

from cerbos_amd.engine import Conf, HipEvaluator
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.lower.celc import LoweringError
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies
from helpers import norm_actions
from oracle.check import EvalParams, RuleTableOracle
from test_fuzz_parity import NOW, _policies, _requests

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("CBH_GPU_FUZZ_SEEDS", "24"))


@pytest.mark.parametrize("seed", range(N_SEEDS))
@pytest.mark.parametrize("pool", ["base", "wide"])
def test_fuzz_store_on_gpu(seed, pool):
    WIDE = pool == "wide"
    rng = np.random.default_rng((10_000 if not WIDE else 20_000) + seed)
    rt = rule_table_from_policies(policies_from_docs(_policies(rng, wide=WIDE)))
    try:
        lt = lower_rule_table(rt)
    except LoweringError:
        pytest.skip("store refused by the lowering")
    ev = HipEvaluator(lt, Conf())
    orc = RuleTableOracle(rt)
    inputs = _requests(rng, 150, wide=WIDE)
    compared = 0
    for lenient, strict in ((False, False), (True, False), (False, True)):
        outs, bad = ev.check(inputs, now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict, allow_unsupported=True)
        params = EvalParams(now_ns=NOW, lenient_scope_search=lenient, strict_evaluation=strict)
        for i, (inp, have) in enumerate(zip(inputs, outs)):
            if i in bad:
                continue
            want = orc.check(inp, params)
            assert norm_actions(have) == norm_actions(want), (seed, lenient, strict, inp)
            assert sorted(have["effectiveDerivedRoles"]) == sorted(want.get("effectiveDerivedRoles") or []), (seed, inp)
            compared += 1
    ev.close()
    assert compared > 150
