"""``python -m cerbos_amd.lower`` - the lowering as a command a non-Python host can run.

A Go host holds its compiled policies as ``runtimev1.RuleTable`` (``internal/ruletable/ruletable.go:637-691``; the manager
rebuilds it on every storage event, ``manager.go:86-124``).  It marshals that message (``proto.Marshal``), hands the bytes
to this command and loads what comes back with ``cbh_table_load`` / ``cbi_table_open``:

    python -m cerbos_amd.lower ruletable.pb image.cbh [--globals '{"environment": "prod"}' | --per-call-globals] [--no-trace] [--stats]
    some-producer | python -m cerbos_amd.lower - - > image.cbh         # stdin -> stdout
    python -m cerbos_amd.lower --policies ./policies image.cbh         # from a policy directory (YAML), for tools and tests

Exit status 0 and the image on success; 2 and a message on stderr when the table cannot be lowered (``LoweringError``:
constructs whose reference behaviour depends on evaluation history - the caller keeps such a table on its CPU engine).
``--stats`` prints one JSON line on stderr: sizes, which decision kernels the table is eligible for, what is outside the
device subset (requests that reach those expressions come back CBH_ST_UNSUPPORTED, never a guessed effect).
"""
from __future__ import annotations

import argparse
import json
import sys


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m cerbos_amd.lower", description="runtimev1.RuleTable bytes -> device table image")
    ap.add_argument("source", help="serialized runtimev1.RuleTable ('-' = stdin), or with --policies a directory of policy YAML")
    ap.add_argument("image", help="where to write the image ('-' = stdout)")
    ap.add_argument("--policies", action="store_true", help="SOURCE is a policy directory, compiled with the bundled front-end")
    ap.add_argument("--globals", default=None, help="JSON object: the engine's configured globals (evaluator/conf.go:40), constants of the image")
    ap.add_argument("--per-call-globals", action="store_true",
                    help="do not fold globals into the image: `G.x` is read from the globals every call brings (cbh_wire_flatten / cbi_flatten_pb_g "
                         "globals_pb: evaluator.EvalParams.Globals) - one image for any globals")
    ap.add_argument("--no-trace", action="store_true", help="leave the trace pass's sections out (decisions only: no evaluation_errors / outputs)")
    ap.add_argument("--stats", action="store_true", help="print the lowering's statistics as one JSON line on stderr")
    args = ap.parse_args(argv)

    from .blob import lower_rule_table
    from .celc import LoweringError
    globals_ = json.loads(args.globals) if args.globals else None
    if globals_ is not None and not isinstance(globals_, dict):
        print("--globals must be a JSON object", file=sys.stderr)
        return 2
    try:
        if args.policies:
            from ..policy.loader import load_policy_dir_with_sources
            from ..ruletable.build import rule_table_from_policies
            rt = rule_table_from_policies(*load_policy_dir_with_sources(args.source), require_ancestors=True)
        else:
            from ..ruletable.proto import decode_rule_table
            wire = sys.stdin.buffer.read() if args.source == "-" else open(args.source, "rb").read()
            rt = decode_rule_table(wire)
        lt = lower_rule_table(rt, globals_, trace=not args.no_trace, per_call_globals=args.per_call_globals)
    except LoweringError as e:
        print("cannot lower this rule table: %s" % e, file=sys.stderr)
        return 2
    except (ValueError, KeyError, OSError) as e:
        print("bad input: %s" % e, file=sys.stderr)
        return 2
    if args.image == "-":
        sys.stdout.buffer.write(lt.blob)
    else:
        with open(args.image, "wb") as fh:
            fh.write(lt.blob)
    if args.stats:
        print(json.dumps(stats_of(lt), sort_keys=True), file=sys.stderr)
    return 0


def stats_of(lt) -> dict:
    st = dict(lt.stats)
    st["unsupported"] = [list(x) for x in lt.unsupported]
    return st


if __name__ == "__main__":
    sys.exit(main())
