"""The shipped library's prologues, read from its code object (no GPU): a decision kernel's wave may wait for everything it has in
flight at most three times between its entry and its first record in the static order of the code - the request words with whatever
depends on nothing, the role ids, the columns' copies (round 5's library had five such waits, each a round trip to
memory).  What this guards is not a result but a property of the COMPILED code that the source does not show: a load moved into a
conditional block, an LDS store placed behind the asynchronous copies, a load and its LDS store in one loop body bring the extra
waits back without failing any other test."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cerbos_amd", "libcerbos_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")) or shutil.which("objcopy") is None,
                    reason="needs the built library and the ROCm LLVM tools")
def test_the_decision_kernels_wait_for_everything_at_most_three_times_before_their_first_record():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_prologue_waits.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    for k in ("cbh_check_flat_kernel:", "cbh_check_flat_kernel_masks:", "cbh_walk2_kernel:"):
        assert k in r.stdout, r.stdout[-2000:]
