import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _sim_engine_if_asked():
    """CBH_TEST_SIM_ENGINE=1 (a developer's aid, never the driver's runs): the GPU tier's test bodies against the simulator build of the
    library (tests/sim_engine.py) - `CBH_TEST_SIM_ENGINE=1 pytest tests -m gpu -k ...` on a machine without a GPU.  Slow at the tier's
    full sizes; proves the host logic and the kernels' source, nothing about the hardware."""
    if os.environ.get("CBH_TEST_SIM_ENGINE") != "1":
        yield
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sim_engine import sim_engine
    with sim_engine():
        yield
