"""CPU tier: the C-ABI library loads and exports every symbol include/cerbos_hip.h declares;
without a GPU the product path fails loudly instead of falling back."""
import ctypes
import os
import re

import pytest

from cerbos_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "cerbos_hip.h")).read()
    return sorted(set(re.findall(r"\b(cbh_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = ctypes.CDLL(capi.LIB_PATH)
    declared = _declared()
    assert sorted(capi.EXPORTED_SYMBOLS) == declared
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.HipEngineError):
        capi.init(0)


def test_ingest_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from cerbos_amd import ingest
    text = open(os.path.join(ROOT, "include", "cerbos_ingest.h")).read()
    declared = sorted(set(re.findall(r"\b(cbi_[a-z_]+)\s*\(", text)))
    assert len(declared) == 23
    lib = ctypes.CDLL(ingest.LIB_PATH)
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_lower_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build_lower()
    text = open(os.path.join(ROOT, "include", "cerbos_lower.h")).read()
    declared = sorted(set(re.findall(r"\b(cbl_[a-z_]+)\s*\(", text)))
    assert declared == ["cbl_abi_version", "cbl_free", "cbl_last_stats_json", "cbl_lower_ruletable_pb", "cbl_lower_ruletable_pb_stats", "cbl_planner_close", "cbl_planner_open",
                        "cbl_planner_plan_pb"]
    lib = ctypes.CDLL(os.path.join(ROOT, "cerbos_amd", "libcerbos_lower.so"))
    for sym in declared:
        assert getattr(lib, sym) is not None
