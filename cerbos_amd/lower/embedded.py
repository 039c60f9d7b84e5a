"""What libcerbos_lower.so (include/cerbos_lower.h, csrc/cbl_lower.cpp) calls inside the interpreter it embeds."""
from __future__ import annotations

import json

OK, CANNOT_LOWER, BAD_INPUT = 0, 2, 3
PER_CALL_GLOBALS, NO_TRACE = 1, 2


def lower_pb(wire: bytes, globals_json, flags: int):
    """-> (status, image bytes or message, statistics JSON)"""
    from ..ruletable.proto import decode_rule_table
    from .blob import lower_rule_table
    from .celc import LoweringError
    try:
        globals_ = json.loads(globals_json) if globals_json else None
        if globals_ is not None and not isinstance(globals_, dict):
            return BAD_INPUT, "globals must be a JSON object", ""
        rt = decode_rule_table(wire)
    except (ValueError, KeyError) as e:
        return BAD_INPUT, "bad input: %s" % e, ""
    try:
        lt = lower_rule_table(rt, globals_, trace=not flags & NO_TRACE, per_call_globals=bool(flags & PER_CALL_GLOBALS))
    except LoweringError as e:
        return CANNOT_LOWER, "cannot lower this rule table: %s" % e, ""
    except (ValueError, KeyError) as e:
        return BAD_INPUT, "bad input: %s" % e, ""
    from .__main__ import stats_of
    return OK, bytes(lt.blob), json.dumps(stats_of(lt), sort_keys=True)
