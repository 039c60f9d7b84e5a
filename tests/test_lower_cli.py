"""``python -m cerbos_amd.lower``: serialized runtimev1.RuleTable in, table image out - the route a Go host takes
(INTEGRATION.md §1): the image must be byte for byte what the in-process lowering of the same table gives, and the C++
ingest and the kernel source (host simulator) must accept it."""
import json
import os
import subprocess
import sys


from cerbos_amd.lower.blob import lower_rule_table
from cerbos_amd.ruletable.proto import encode_rule_table
from helpers import store_rule_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLOBALS = {"environment": "test"}


def _run(args, stdin=None):
    return subprocess.run([sys.executable, "-m", "cerbos_amd.lower"] + args, input=stdin, capture_output=True, cwd=ROOT,
                          env=dict(os.environ, PYTHONPATH=ROOT))


def test_image_from_rule_table_bytes_is_the_in_process_image(tmp_path):
    rt = store_rule_table()
    wire = encode_rule_table(rt)
    src, dst = tmp_path / "ruletable.pb", tmp_path / "image.cbh"
    src.write_bytes(wire)
    p = _run([str(src), str(dst), "--globals", json.dumps(GLOBALS), "--stats"])
    assert p.returncode == 0, p.stderr.decode()
    want = lower_rule_table(rt, GLOBALS).blob
    assert dst.read_bytes() == want
    stats = json.loads(p.stderr.decode().strip().splitlines()[-1])
    assert stats["walk2"] is True and stats["rows"] > 50
    # stdin -> stdout
    q = _run(["-", "-", "--globals", json.dumps(GLOBALS)], stdin=wire)
    assert q.returncode == 0 and q.stdout == want
    # the consumers of an image take it
    from cerbos_amd.ingest import IngestTable
    it = IngestTable(dst.read_bytes())
    it.close()


def test_bad_input_is_an_error_not_an_image(tmp_path):
    dst = tmp_path / "image.cbh"
    p = _run(["-", str(dst)], stdin=b"\x0a\xff\xff\xff\xff\x0f")
    assert p.returncode == 2 and not dst.exists() and p.stderr
    p = _run(["-", str(dst), "--globals", "[1]"], stdin=b"")
    assert p.returncode == 2


def test_per_call_globals_flag(tmp_path):
    """--per-call-globals: the image reads `G.x` from the call (a column of root "G"), byte for byte the in-process lowering"""
    rt = store_rule_table()
    src, dst = tmp_path / "ruletable.pb", tmp_path / "image.cbh"
    src.write_bytes(encode_rule_table(rt))
    p = _run([str(src), str(dst), "--per-call-globals"])
    assert p.returncode == 0, p.stderr.decode()
    lt = lower_rule_table(rt, per_call_globals=True)
    assert dst.read_bytes() == lt.blob and any(root == "G" for root, _ in lt.columns)
