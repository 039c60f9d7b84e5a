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


# ---- PlanResources behind the same library (include/cerbos_lower.h cbl_planner_*): a planner per published rule table
_PLANNERS = {}
_NEXT = [1]


def planner_open(wire: bytes):
    """-> (status, handle or message)"""
    from ..plan import Planner
    from ..ruletable.proto import decode_rule_table
    try:
        rt = decode_rule_table(wire)
    except (ValueError, KeyError) as e:
        return BAD_INPUT, "bad input: %s" % e
    h = _NEXT[0]
    _NEXT[0] += 1
    _PLANNERS[h] = Planner(rt)
    return OK, h


def planner_close(handle: int):
    _PLANNERS.pop(handle, None)
    return OK


def planner_plan_pb(handle: int, input_pb: bytes, params_json):
    """One serialized enginev1.PlanResourcesInput -> (status, serialized PlanResourcesOutput or message).  params_json: the call's
    evaluator parameters {"globals": {..}, "defaultPolicyVersion", "defaultScope", "lenientScopeSearch", "strictEvaluation", "nowNs"}"""
    from .. import wire as _wire
    p = _PLANNERS.get(handle)
    if p is None:
        return BAD_INPUT, "unknown planner handle"
    try:
        params = json.loads(params_json) if params_json else {}
        inp = _wire.decode_plan_resources_input(input_pb)
    except (ValueError, KeyError, IndexError) as e:
        return BAD_INPUT, "bad input: %s" % e
    out = p.plan(inp, globals_=params.get("globals"), default_policy_version=params.get("defaultPolicyVersion") or "default",
                 default_scope=params.get("defaultScope") or "", lenient_scope_search=bool(params.get("lenientScopeSearch")),
                 strict_evaluation=bool(params.get("strictEvaluation")), now_ns=params.get("nowNs"))
    return OK, _wire.encode_plan_resources_output(out)
