"""Host-side mirror of the reference's evaluator seam for the CheckResources hot path.

Reference interface (``internal/evaluator/evaluator.go:16-19``)::

    type Evaluator interface {
        Check(ctx, []*enginev1.CheckInput, ...CheckOpt) ([]*enginev1.CheckOutput, error)
        ...
    }

implemented by ``engine.(*Engine).Check`` (``internal/engine/engine.go:216-240``).
``HipEvaluator.check`` takes the same inputs (JSON-shaped ``CheckInput`` dicts), the same
per-call options (``CheckOpt`` -> keyword arguments: ``evaluator.go:22-73``) and returns
JSON-shaped ``CheckOutput`` dicts (``check.go:64-94``).  All decisions are made by the HIP
kernels behind the C ABI (``capi``); nothing here evaluates policies on the CPU.
"""
from __future__ import annotations

import threading
import time

import numpy as np

from . import capi, namer
from .flatten import Flattener
from .lower.blob import LoweredTable, lower_rule_table
from .policy.loader import load_policy_dir_with_sources, policies_from_docs
from .ruletable.build import rule_table_from_policies

_EFFECT_NAMES = {capi.EFFECT_ALLOW: "EFFECT_ALLOW", capi.EFFECT_DENY: "EFFECT_DENY"}


class DeviceUnsupported(RuntimeError):
    """Some inputs hit an operation outside the device subset (status CBH_ST_UNSUPPORTED).
    The caller (in production: ``ruletable.Manager.Check``) must evaluate those inputs with
    its own engine; this package never does so itself."""

    def __init__(self, request_indices, expressions):
        self.request_indices = request_indices
        self.expressions = expressions
        super().__init__(
            "%d input(s) need CEL features outside the device subset (%s)"
            % (len(request_indices), "; ".join(e for e, _ in expressions[:3]) or "runtime limits"))


class Conf:
    """evaluator.Conf (``internal/evaluator/conf.go:38-67``) - the fields the path reads."""

    def __init__(self, default_policy_version="default", default_scope="", globals_=None,
                 lenient_scope_search=False, strict_evaluation=False):
        self.default_policy_version = default_policy_version
        self.default_scope = default_scope
        self.globals = dict(globals_ or {})
        self.lenient_scope_search = lenient_scope_search
        self.strict_evaluation = strict_evaluation


def effective_policy_keys(policy_keys, mask_words):
    """A row of ``cbh_check_batch_trail``'s masks -> the keys of AuditTrail.EffectivePolicies: the policies whose bit is set and, for a
    scoped resource / principal policy, its ancestors among the table's policies - the source attributes a policy SET carries
    (compile.go:153-180, 474-494; a role policy carries its own only, compile.go:116-117)."""
    have = set(policy_keys)
    out = set()
    for i, key in enumerate(policy_keys):
        if not (int(mask_words[i >> 5]) >> (i & 31)) & 1:
            continue
        out.add(key)
        if "/" in key and not key.startswith("role."):
            head, scope = key.split("/", 1)
            while scope:
                scope = scope.rpartition(".")[0]
                anc = head + "/" + scope if scope else head
                if anc in have:
                    out.add(anc)
    return sorted(out)


class HipEvaluator:
    _ingest = None
    _py_flattener = None
    _ingest_lock = threading.Lock()   # check_pb / check_request_pb are called from many threads

    def __init__(self, lowered: LoweredTable, conf: Conf = None, device: int = 0, native_ingest: bool = False):
        """``native_ingest``: flatten through the protobuf wire format and libcerbos_ingest.so (what a Go
        caller does, include/cerbos_ingest.h) instead of the Python flattener; same batch either way."""
        self.conf = conf or Conf()
        self.lt = lowered
        if capi._inited_device is None:
            capi.init(device)
        self.table = capi.Table(lowered.blob)
        if native_ingest:
            from .ingest import WireFlattener
            self.flattener = WireFlattener(lowered)
        else:
            self.flattener = Flattener(lowered)

    # -- constructors ---------------------------------------------------------------------
    @classmethod
    def from_rule_table(cls, rt: dict, conf: Conf = None, device: int = 0, per_call_globals: bool = False):
        """``per_call_globals``: the image does not fold ``conf.globals`` in; every ``check`` brings its globals
        (``EvalParams.Globals``, evaluator.go:52-57; default: ``conf.globals``) and one image answers any of them."""
        conf = conf or Conf()
        ev = cls(lower_rule_table(rt, conf.globals, per_call_globals=per_call_globals), conf, device)
        ev.rt = rt          # the query planner works on the rule table itself (plan_resources)
        return ev

    @classmethod
    def from_rule_table_pb(cls, wire: bytes, conf: Conf = None, device: int = 0):
        """From the reference's own artefact: serialized ``runtimev1.RuleTable`` (what ``ruletable.Manager`` holds,
        ``private/ruletable/ruletable.go:27-44``) - no policy YAML, no compile step on this side."""
        from .ruletable.proto import decode_rule_table
        return cls.from_rule_table(decode_rule_table(wire), conf, device)

    @classmethod
    def from_policies(cls, docs, conf: Conf = None, device: int = 0):
        return cls.from_rule_table(rule_table_from_policies(policies_from_docs(docs)), conf, device)

    @classmethod
    def from_policy_dir(cls, path: str, conf: Conf = None, device: int = 0):
        policies, sources = load_policy_dir_with_sources(path)   # compile errors carry file:line:column; scope ancestors must exist
        return cls.from_rule_table(rule_table_from_policies(policies, sources, require_ancestors=True), conf, device)

    # -- the seam -------------------------------------------------------------------------
    def _call_globals(self, globals_):
        """The globals of one call: the override, else the configured ones; None for a table that carries them in its image."""
        if not getattr(self.lt, "per_call_globals", False):
            if globals_ is not None and dict(globals_) != self.conf.globals:
                raise ValueError("this table was lowered with its globals folded in: lower it with per_call_globals=True to override them per call")
            return None
        return dict(self.conf.globals if globals_ is None else globals_)

    def check(self, inputs, now_ns=None, lenient_scope_search=None, strict_evaluation=None,
              default_policy_version=None, default_scope=None, allow_unsupported=False, trace=False, globals_=None):
        """``Evaluator.Check``: one CheckOutput per CheckInput, same order.

        ``trace``: also fill ``evaluationErrors`` and ``outputs`` as check.go:90-92 does.  They come from a second launch
        (``cbh_trace_batch``) over the inputs that can have any: the ones a decision kernel marked CBH_ST_CEL_ERROR, or all
        of them when the table has variables (evaluated whether a condition reads them or not) or output expressions.
        With ``allow_unsupported`` the result is then (outputs, unsupported inputs, {input: {"errors", "outputs"}} = what the
        device could not name all of for that input); without it such inputs raise."""
        conf = self.conf
        lenient = conf.lenient_scope_search if lenient_scope_search is None else lenient_scope_search
        strict = conf.strict_evaluation if strict_evaluation is None else strict_evaluation
        dver = conf.default_policy_version if default_policy_version is None else default_policy_version
        dscope = conf.default_scope if default_scope is None else default_scope
        if now_ns is None:
            now_ns = time.time_ns()  # frozen once per call (evaluator_trace_common.go:24-26)
        g = self._call_globals(globals_)
        batch = self.flattener.flatten(inputs, dver, dscope) if g is None else self.flattener.flatten(inputs, dver, dscope, globals_=g)
        flags = capi.F_WANT_DERIVED_ROLES
        if lenient:
            flags |= capi.F_LENIENT_SCOPE_SEARCH
        if strict:
            flags |= capi.F_STRICT_EVALUATION
        res = self.table.check(batch, now_ns=now_ns, flags=flags)
        if not trace:
            return self.assemble(inputs, batch, res, dver, allow_unsupported)
        outs, bad = self.assemble(inputs, batch, res, dver, True)
        incomplete = self._trace(inputs, batch, res, outs, bad, now_ns, flags, dver, dscope, g)
        if (bad or incomplete) and not allow_unsupported:
            raise DeviceUnsupported(sorted(set(bad) | set(incomplete)), self.lt.unsupported + self.lt.trace_unsupported)
        return (outs, bad, incomplete) if allow_unsupported else outs

    def _trace(self, inputs, batch, res, outs, bad, now_ns, flags, dver, dscope, g=None):
        """The trace pass for the inputs that need it; fills outs[i]["evaluationErrors"] / ["outputs"] in place and returns
        the inputs left incomplete."""
        from .trace import TraceDecoder
        lt = self.lt
        for o in outs:
            o["evaluationErrors"] = []
            o["outputs"] = []
        # the inputs with a tuple marked CBH_ST_CEL_ERROR or CBH_ST_WANTS_TRACE (cerbos_hip.h): the walk marks what the trace
        # pass has something for - an absorbed error, a visited rule with outputs, a variable it could not evaluate; a kernel
        # that cannot tell marks every tuple of a table with variables or outputs
        sel, t = [], 0
        skip = set(bad)
        for i in range(len(inputs)):
            n = len(batch.actions_per_request[i])
            if i not in skip and np.isin(res.status[t:t + n], (capi.ST_CEL_ERROR, capi.ST_WANTS_TRACE)).any():
                sel.append(i)
            t += n
        self.last_traced = len(sel)
        if not sel:
            return {}
        sub = [inputs[i] for i in sel]
        if self._py_flattener is None:
            with self._ingest_lock:   # check() is called from many threads
                if self._py_flattener is None:
                    self._py_flattener = Flattener(lt)
        sbatch = self._py_flattener.flatten(sub, dver, dscope, globals_=g)
        tres, records = self.table.trace(sbatch, now_ns=now_ns, flags=flags)
        decoded = TraceDecoder(lt, sbatch, sub, g).decode(records, len(records), tres.status)
        # the tracing kernel decides the inputs again: anything but the same effects is a defect, never to be papered over
        tin = tres.to_input_order(sbatch)
        t = 0
        for j, i in enumerate(sel):
            for a in sbatch.actions_per_request[j]:
                if _EFFECT_NAMES[int(tin.effect[t])] != outs[i]["actions"][a]["effect"] and int(tin.status[t]) != capi.ST_UNSUPPORTED:
                    raise RuntimeError("trace pass and decision pass disagree on input %d action %r" % (i, a))
                t += 1
        incomplete = {}
        for j, i in enumerate(sel):
            d = decoded[j]
            outs[i]["evaluationErrors"] = d["evaluationErrors"]
            outs[i]["outputs"] = d["outputs"]
            what = set(d["incomplete"])
            if what:
                incomplete[i] = what
        return incomplete

    def check_pb(self, data, offsets, now_ns=None, lenient_scope_search=None, strict_evaluation=None,
                 default_policy_version=None, default_scope=None, trace=False, device_ingest=True, globals_=None):
        """Bytes in, bytes out - the path a Go caller takes: serialized ``CheckInput``
        messages (``data`` uint8, ``offsets`` uint64[n + 1]) -> serialized ``CheckOutput`` messages.
        ``device_ingest`` (the default): the device road - the raw bytes cross PCIe, the GPU flattens them
        (``cbh_wire_flatten``), decides and writes the answers (``cbh_wire_outputs``); messages the device flattener
        leaves to the host (``capi.HostFlattenerNeeded``) and ``device_ingest=False`` take the host road: C++ ingest ->
        ``cbh_check_batch`` -> C++ response assembly.  Same bytes either way (tests/test_gpu_wire.py).
        Returns ([serialized CheckOutput], flags uint8[n]); flags bit 0 = the device could not
        evaluate that input (the caller's own engine must), bit 1 = a CEL error was absorbed."""
        conf = self.conf
        lenient = conf.lenient_scope_search if lenient_scope_search is None else lenient_scope_search
        strict = conf.strict_evaluation if strict_evaluation is None else strict_evaluation
        dver = conf.default_policy_version if default_policy_version is None else default_policy_version
        dscope = conf.default_scope if default_scope is None else default_scope
        if now_ns is None:
            now_ns = time.time_ns()
        self._ingest_table()
        g = self._call_globals(globals_)
        gpb = b""
        if g is not None:
            from . import wire as _wire
            gpb = _wire.encode_map(1, g)   # google.protobuf.Struct: fields = 1
        flags = capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0) | (capi.F_STRICT_EVALUATION if strict else 0)
        outs = None
        self.last_road = "host"
        if device_ingest:
            try:
                db = self.table.wire_flatten(data, offsets, dver, dscope, globals_pb=gpb)
            except capi.HostFlattenerNeeded:
                db = None
            if db is not None:
                try:
                    self.table.launch(db, now_ns=now_ns, flags=flags)
                    outs, oflags = self.table.wire_outputs(db)
                    self.last_road = "device"
                finally:
                    db.close()
        if outs is None:
            batch = self._ingest.flatten_pb(data, offsets, dver, dscope, globals_pb=gpb)
            res = self.table.check(batch, now_ns=now_ns, flags=flags, device_order=True)
            outs, oflags = self._ingest.assemble_pb(batch, res, data, offsets, dver)
        if not trace:
            return outs, oflags
        # evaluation_errors / outputs (check.go:90-92): the inputs that can have any go through the tracing kernel and
        # cbi_trace_pb; its bytes - just those two fields - are appended to the CheckOutput (protobuf concatenation merges).
        # flags bits 2 / 3 (CBI_TRACE_*): the device could not name all errors / outputs of that input.
        import numpy as np

        from .ingest import trace_pb
        oflags = np.array(oflags, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sel = [i for i in range(len(outs)) if not (oflags[i] & 1) and (oflags[i] & (2 | 16))]   # CBI_OUT_CEL_ERROR | CBI_OUT_WANTS_TRACE
        if not sel:
            return outs, oflags
        data = np.ascontiguousarray(data, dtype=np.uint8)
        parts = [data[int(offsets[i]):int(offsets[i + 1])] for i in sel]
        sdata = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
        soff = np.zeros(len(sel) + 1, dtype=np.uint64)
        soff[1:] = np.cumsum([p.size for p in parts])
        sbatch = self._ingest.flatten_pb(sdata, soff, dver, dscope, globals_pb=gpb)
        tres, records = self.table.trace(sbatch, now_ns=now_ns, flags=flags)
        extra, tflags = trace_pb(self._ingest, sbatch, tres, records, sdata, soff)
        for j, i in enumerate(sel):
            outs[i] = outs[i] + extra[j]
            oflags[i] = (int(oflags[i]) & 0xEF) | (int(tflags[j]) & 12)   # traced: CBI_OUT_WANTS_TRACE is answered
        return outs, oflags

    def check_request_pb(self, request: bytes, aux_data: bytes = None, now_ns=None, lenient_scope_search=None,
                         strict_evaluation=None, default_policy_version=None, default_scope=None, trace=False):
        """``svc.CheckResources`` on bytes (cerbos_svc.go:255-344): one serialized ``CheckResourcesRequest`` (and the
        serialized engine ``AuxData`` derived from its JWT) -> (serialized ``CheckResourcesResponse``, flags per
        resource entry: bit 0 = the caller's own engine must evaluate that entry)."""
        conf = self.conf
        lenient = conf.lenient_scope_search if lenient_scope_search is None else lenient_scope_search
        strict = conf.strict_evaluation if strict_evaluation is None else strict_evaluation
        dver = conf.default_policy_version if default_policy_version is None else default_policy_version
        dscope = conf.default_scope if default_scope is None else default_scope
        if now_ns is None:
            now_ns = time.time_ns()
        self._ingest_table()
        batch = self._ingest.flatten_request_pb(request, aux_data, dver, dscope)
        flags = capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0) | (capi.F_STRICT_EVALUATION if strict else 0)
        res = self.table.check(batch, now_ns=now_ns, flags=flags, device_order=True)
        traced = None
        if trace and self._ingest.trace_scope() == 2:
            # a table with output expressions: the entries go through the tracing kernel too (the same batch) and their
            # outputs into ResultEntry.outputs (cerbos_svc.go:325-327).  (The response carries no evaluation errors.)
            traced = self.table.trace(batch, now_ns=now_ns, flags=flags)
        raw, oflags = self._ingest.assemble_response_pb(batch, res, request, dver, traced=traced, aux_data=aux_data)
        if traced is not None:
            oflags = oflags & 0xEF   # traced: CBI_OUT_WANTS_TRACE is answered
        return raw, oflags

    def check_requests_pb(self, requests, aux=None, now_ns=None, lenient_scope_search=None, strict_evaluation=None,
                          default_policy_version=None, default_scope=None, audit_trail=False):
        """Many serialized ``CheckResourcesRequest``s at once down the device road (``cbh_wire_check_requests_pb``: the requests are
        split into the ``CheckInput``s of cerbos_svc.go:274-288 on the device) -> ([[serialized CheckOutput] per request], flags per
        input, include_meta per request).  ``aux``: per request the serialized engine ``AuxData`` or None.  Requests the device road
        leaves to the host flattener take ``check_request_pb`` one by one (``capi.HostFlattenerNeeded``).  ``audit_trail``: a fourth
        value, per request the sorted keys of AuditTrail.EffectivePolicies (``cbh_wire_check_requests_trail_pb``: what the one
        decision-log entry of the call carries, check.go:302-304)."""
        conf = self.conf
        lenient = conf.lenient_scope_search if lenient_scope_search is None else lenient_scope_search
        strict = conf.strict_evaluation if strict_evaluation is None else strict_evaluation
        dver = conf.default_policy_version if default_policy_version is None else default_policy_version
        dscope = conf.default_scope if default_scope is None else default_scope
        if now_ns is None:
            now_ns = time.time_ns()
        flags = capi.F_WANT_DERIVED_ROLES | (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0) | (capi.F_STRICT_EVALUATION if strict else 0)
        if not audit_trail:
            return self.table.wire_check_requests_pb(requests, aux, now_ns=now_ns, flags=flags, default_policy_version=dver, default_scope=dscope)
        outs, oflags, meta, masks = self.table.wire_check_requests_pb(requests, aux, now_ns=now_ns, flags=flags, default_policy_version=dver,
                                                                      default_scope=dscope, trail=True)
        keys = self.lt.policy_keys
        return outs, oflags, meta, [effective_policy_keys(keys, row) for row in masks]

    def effective_policies(self, inputs, now_ns=None, lenient_scope_search=None, strict_evaluation=None, default_policy_version=None,
                           default_scope=None, globals_=None, per_input=False):
        """``engine.Check``'s second return value for ``inputs`` as ONE call: the keys of AuditTrail.EffectivePolicies
        (engine.go:289-338, check.go:302-304) - sorted; ``per_input``: one list per input instead of the call's union.
        (``cbh_check_batch_trail``: the general walk decides the batch once more, in the reference's order of roles.)"""
        conf = self.conf
        lenient = conf.lenient_scope_search if lenient_scope_search is None else lenient_scope_search
        strict = conf.strict_evaluation if strict_evaluation is None else strict_evaluation
        dver = conf.default_policy_version if default_policy_version is None else default_policy_version
        dscope = conf.default_scope if default_scope is None else default_scope
        if now_ns is None:
            now_ns = time.time_ns()
        g = self._call_globals(globals_)
        batch = self.flattener.flatten(inputs, dver, dscope) if g is None else self.flattener.flatten(inputs, dver, dscope, globals_=g)
        flags = (capi.F_LENIENT_SCOPE_SEARCH if lenient else 0) | (capi.F_STRICT_EVALUATION if strict else 0)
        n = len(inputs)
        groups = None
        if per_input:   # device request i <- pre-sort (virtual) request <- the CheckInput it came from
            pre = np.arange(batch.n_requests) if batch.req_perm is None else np.asarray(batch.req_perm)
            groups = np.asarray(batch.vreq_input)[pre].astype(np.uint32)
        _, masks = self._check_trail(batch, groups, n if per_input else 1, now_ns, flags)
        keys = [effective_policy_keys(self.lt.policy_keys, row) for row in masks]
        return keys if per_input else keys[0]

    def plan_resources(self, inp, now_ns=None, lenient_scope_search=None, strict_evaluation=None, default_policy_version=None,
                       default_scope=None, globals_=None):
        """``Engine.PlanResources`` (engine.go:141-170, ruletable/plan.go): a ``PlanResourcesInput`` (JSON shape) -> the
        ``PlanResourcesOutput`` - filter {kind, condition}, filterDebug, matchedScopes, evaluationErrors - and, beside it, the call's
        ``effectivePolicies``.  Host-side and symbolic (cerbos_amd/plan), as in the reference: the GPU is not involved."""
        rt = getattr(self, "rt", None)
        if rt is None:
            raise ValueError("plan_resources needs the rule table: build the evaluator with from_rule_table / from_policies / from_policy_dir")
        if getattr(self, "_planner", None) is None:
            from .plan import Planner
            self._planner = Planner(rt)
        conf = self.conf
        return self._planner.plan(
            inp, globals_=dict(conf.globals if globals_ is None else globals_),
            default_policy_version=conf.default_policy_version if default_policy_version is None else default_policy_version,
            default_scope=conf.default_scope if default_scope is None else default_scope,
            lenient_scope_search=conf.lenient_scope_search if lenient_scope_search is None else lenient_scope_search,
            strict_evaluation=conf.strict_evaluation if strict_evaluation is None else strict_evaluation,
            now_ns=time.time_ns() if now_ns is None else now_ns)

    def _check_trail(self, batch, groups, n_groups, now_ns, flags):
        return self.table.check_trail(batch, groups, n_groups, now_ns=now_ns, flags=flags)

    def _ingest_table(self):
        if self._ingest is None:
            with self._ingest_lock:
                if self._ingest is None:
                    from .ingest import IngestTable
                    self._ingest = IngestTable(self.lt.blob)
        return self._ingest

    def assemble(self, inputs, batch, res, default_policy_version, allow_unsupported=False):
        """ids -> CheckOutput (check.go:64-94)."""
        lt = self.lt
        outs = []
        bad = []
        t = 0
        for r, inp in enumerate(inputs):
            actions = {}
            unsupported = False
            for a in batch.actions_per_request[r]:
                st = int(res.status[t]) if res.status is not None else 0
                if st == capi.ST_UNSUPPORTED:
                    unsupported = True
                eff = _EFFECT_NAMES[int(res.effect[t])]
                pol = self._policy_string(int(res.policy[t]), inp, default_policy_version)
                sc = int(res.scope[t])
                scope = "" if sc == capi.NONE else lt.scopes[sc]
                prev = actions.get(a)
                if prev is None or eff == "EFFECT_DENY" or prev["effect"] != "EFFECT_DENY":
                    actions[a] = {"effect": eff, "policy": pol, "scope": scope}  # setEffect: DENY sticky
                t += 1
            if unsupported:
                bad.append(r)
            mask = int(res.edr[r]) if res.edr is not None else 0
            edr = [n for i, n in enumerate(lt.dr_names) if (mask >> i) & 1]
            outs.append({
                "requestId": inp.get("requestId", ""),
                "resourceId": inp["resource"].get("id", ""),
                "actions": actions,
                "effectiveDerivedRoles": edr,
            })
        if bad and not allow_unsupported:
            raise DeviceUnsupported(bad, lt.unsupported)
        if allow_unsupported:
            return outs, bad
        return outs

    def _policy_string(self, word, inp, default_policy_version):
        kind, ident = word >> 28, word & 0x0FFFFFFF
        if kind == capi.P_EMPTY:
            return ""
        if kind == capi.P_NO_MATCH:
            return "NO_MATCH"
        if kind == capi.P_NO_MATCH_SP:
            return "NO_MATCH_FOR_SCOPE_PERMISSIONS"
        if kind == capi.P_TABLE:
            return self.lt.policy_keys[ident]
        scope = self.lt.scopes[ident]
        if kind == capi.P_RESOURCE:
            ver = inp["resource"].get("policyVersion", "") or default_policy_version
            return namer.policy_key_from_fqn(namer.resource_policy_fqn(inp["resource"]["kind"], ver, scope))
        ver = inp["principal"].get("policyVersion", "") or default_policy_version
        return namer.policy_key_from_fqn(namer.principal_policy_fqn(inp["principal"]["id"], ver, scope))

    def close(self):
        if getattr(self, "_ingest", None) is not None:
            self._ingest.close()
            self._ingest = None
        self.table.close()
