"""TEST INFRASTRUCTURE ONLY.  The library's host side (cerbos_amd/csrc/cbh_engine.hip: pools, streams, slices, every entry point's
launches) compiled against tests/hostsim/fakehip (device memory = host memory, a launch = the kernel's source on the fiber scheduler)
-> tests/hostsim/_build/libcerbos_hip_sim.so, and a context manager that makes cerbos_amd.capi talk to it for the duration of ONE
test.  The product never looks for this library; the GPU tier runs the same test bodies against libcerbos_hip.so on the MI355X."""
import contextlib
import os
import subprocess

from cerbos_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cerbos_amd", "csrc")
FAKE = os.path.join(ROOT, "tests", "hostsim", "fakehip")
OUT_DIR = os.path.join(ROOT, "tests", "hostsim", "_build")
LIB = os.path.join(OUT_DIR, "libcerbos_hip_sim.so")


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    return deps + [os.path.join(FAKE, "hip", f) for f in os.listdir(os.path.join(FAKE, "hip"))] + [os.path.join(ROOT, "include", "cerbos_hip.h")]


def build_four_waves():
    """The same with FOUR waves to a workgroup, as the flat and walk kernels have on the GPU (the other simulations run one): the
    per-wave quarters of the dynamic LDS, the class tables the waves share, the barriers between 256 lanes."""
    lib = os.path.join(OUT_DIR, "libcerbos_hip_sim_w4.so")
    if os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(d) for d in _deps()):
        return lib
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = lib + ".%d.tmp" % os.getpid()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-x", "c++", "-fPIC", "-shared", "-DCBH_FLAT_WAVES_OVERRIDE=4u", "-DCBH_HOSTSIM_LDS_WAVES=4",
                           "-I" + FAKE, "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, "cbh_engine.hip"), "-o", tmp, "-lpthread", "-ldl"])
    os.replace(tmp, lib)
    return lib


def build():
    if os.environ.get("CBH_TEST_SIM_LIB"):      # a variant built by hand (e.g. -fsanitize=address: tools/sim_engine_asan.sh)
        return os.environ["CBH_TEST_SIM_LIB"]
    deps = _deps()
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = LIB + ".%d.tmp" % os.getpid()          # (several xdist workers may build at once: each its own file, then an atomic rename)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-x", "c++", "-fPIC", "-shared", "-I" + FAKE, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(CSRC, "cbh_engine.hip"), "-o", tmp, "-lpthread", "-ldl"])
    os.replace(tmp, LIB)
    return LIB


@contextlib.contextmanager
def sim_engine():
    """capi -> the simulator build of the library, then back (the real library stays loaded; its handle returns)."""
    lib_path = build()
    saved = (capi.LIB_PATH, capi._lib, capi._inited_device)
    capi.LIB_PATH, capi._lib, capi._inited_device = lib_path, None, None
    try:
        yield capi
    finally:
        try:
            capi.load().cbh_shutdown()
        except Exception:
            pass
        capi.LIB_PATH, capi._lib, capi._inited_device = saved
