"""GPU tier twin of tests/test_hostsim_cel_kats.py: the reference's CEL known-answer tests
(internal/test/testdata/cel_eval/*.yaml, evaluator_test.go:22-48, frozen now) decided by the kernels on
hardware through the C ABI - global_load_lds, scalar address_space(4) loads, SGPR spills and real wave
divergence are what the host simulator cannot see."""
import pytest

from cerbos_amd.engine import HipEvaluator
from cerbos_amd.lower.celc import LoweringError
import test_hostsim_cel_kats as kats

pytestmark = pytest.mark.gpu


def _gpu(lt, conf):
    return HipEvaluator(lt, conf)


@pytest.mark.parametrize("case", kats.CASES, ids=[c["name"] for c in kats.CASES])
def test_condition_on_gpu(case):
    try:
        got = kats._decide(case, [{"actions": ["kat"], "condition": {"match": case["condition"]}}], _gpu)
    except LoweringError as e:
        pytest.skip("refused by the lowering: %s" % e)
    if got is None:
        pytest.skip("flagged UNSUPPORTED by the device path")
    assert got["kat"] == bool(case["want"]), case["name"]


def test_true_leaves_on_gpu():
    kats.test_true_leaves_on_device_path(_gpu)
