"""The wire walkers of libcerbos_ingest under AddressSanitizer + UndefinedBehaviorSanitizer (the 2-byte
`18 05` attribute value that made value() walk a stale span only crashed under ASan; without it stale spans usually
point at valid memory).  tests/asan/ingest_fuzz.cpp + cbh_ingest.cpp are built with -fsanitize=address,undefined and
run a mutation fuzz over serialized CheckInputs and one CheckResourcesRequest: any out-of-bounds read, use after free
or undefined shift aborts the process and fails the test."""
import os
import subprocess
import sys

import numpy as np
import pytest

from cerbos_amd import wire, workloads
from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.policy.loader import policies_from_docs
from cerbos_amd.ruletable.build import rule_table_from_policies

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz_binary():
    # built in-tree and kept while its sources stand (half a minute of the tier's seven otherwise)
    srcs = [os.path.join(ROOT, "tests", "asan", "ingest_fuzz.cpp"), os.path.join(ROOT, "cerbos_amd", "csrc", "cbh_ingest.cpp"),
            os.path.join(ROOT, "cerbos_amd", "csrc", "cbh_blob.h"), os.path.join(ROOT, "include", "cerbos_ingest.h"), os.path.join(ROOT, "include", "cerbos_hip.h")]
    os.makedirs(os.path.join(ROOT, "tests", "asan", "_build"), exist_ok=True)
    out = os.path.join(ROOT, "tests", "asan", "_build", "ingest_fuzz")
    if os.path.exists(out) and all(os.path.getmtime(x) <= os.path.getmtime(out) for x in srcs):
        return out
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "asan", "ingest_fuzz.cpp"),
           os.path.join(ROOT, "cerbos_amd", "csrc", "cbh_ingest.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime in this toolchain: %s" % r.stderr[-300:])
    return out


@pytest.mark.parametrize("name", ["C2", "C5", "store"])
def test_mutated_wire_input_never_reads_out_of_bounds(fuzz_binary, tmp_path, name):
    if name == "store":   # the golden store: variables, output expressions with templates, role policies (the trace consumer's paths)
        from helpers import load_json, store_rule_table
        lt = lower_rule_table(store_rule_table(), {"environment": "test"})
        inputs = [i for c in load_json("engine_cases.json") for i in c["inputs"]][:64]
    else:
        pol, reqs = {"C2": (workloads.c2_policies, workloads.c2_requests), "C5": (workloads.c5_policies, workloads.c5_requests)}[name]
        lt = lower_rule_table(rule_table_from_policies(policies_from_docs(pol())))
        inputs = reqs(n_requests=64).to_inputs()
    data, off = wire.pack_messages([wire.encode_check_input(i) for i in inputs])
    (tmp_path / "table.blob").write_bytes(lt.blob)
    data.tofile(str(tmp_path / "messages.bin"))
    np.asarray(off, dtype=np.uint64).tofile(str(tmp_path / "offsets.bin"))
    request = {"requestId": "fuzz", "includeMeta": True, "principal": inputs[0]["principal"],
               "resources": [{"actions": i["actions"], "resource": i["resource"]} for i in inputs[:6]]}
    (tmp_path / "request.bin").write_bytes(wire.encode_check_resources_request(request))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_binary, str(tmp_path), "6000", "7"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    ok, refused = (int(x) for x in r.stdout.split()[1::2])
    assert ok > 1000 and refused > 1000, r.stdout   # the fuzz reaches both outcomes
    sys.stdout.write(r.stdout)
