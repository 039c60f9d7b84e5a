#!/usr/bin/env python3
"""One-off wider sweep of the GPU differential fuzz (beyond the seeds of the test tier): flat-kernel stores incl. the
variant with the evaluator call, large buckets and deep chains (tests/test_flat_kernel.py) and general stores with the
wide CEL pool (tests/test_gpu_fuzz.py).   python tools/gpu_fuzz_sweep.py [first_seed] [n_seeds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_flat_kernel as F   # noqa: E402
import test_gpu_fuzz as G      # noqa: E402
from cerbos_amd.engine import Conf, HipEvaluator   # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
budget_s = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # stop cleanly (and say what was done) after this many seconds; 0 = never
mk = lambda lt: HipEvaluator(lt, Conf())   # noqa: E731
t0 = time.time()
done = {"flat": 0, "lists": 0, "large": 0, "deep": 0, "general": 0, "skipped": 0}
last = first - 1
for s in range(first, first + n):
    if budget_s and time.time() - t0 > budget_s:
        break
    last = s
    def flat(kind, **kw):
        # (the generator's own precondition - a store "with lists" must draw a list, one without must not - fails for a rare
        # seed BEFORE anything reaches the device: such a seed is skipped and counted, any other failure stops the sweep)
        import traceback
        try:
            F._run_seed(s, mk, True, **kw); done[kind] += 1
        except AssertionError:
            if "plain != with_lists" in traceback.format_exc().splitlines()[-2]:
                done["skipped"] += 1
            else:
                raise
    flat("flat")
    if s % 3 == 0:
        flat("lists", with_lists=True)
    if s % 5 == 0:
        flat("large", many_rules=True)
    if s % 4 == 0:
        flat("deep", deep=True)
    for pool in ("base", "wide"):
        try:
            G.test_fuzz_store_on_gpu(s, pool); done["general"] += 1
        except BaseException as e:   # pytest.skip raises a BaseException subclass
            if type(e).__name__ == "Skipped":
                done["skipped"] += 1
            else:
                raise
print("sweep clean:", done, "seeds %d..%d" % (first, last), "%.0f s" % (time.time() - t0))
