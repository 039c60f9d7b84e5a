"""Batching across concurrent ``Check`` calls - the GPU analogue of ``engine.checkParallel``.

The reference fans one call's inputs OUT to a worker pool (``internal/engine/engine.go:309-338``); a device wants
the opposite: the inputs of many small concurrent calls gathered IN to one batch.
``BatchingEvaluator.check`` blocks like ``Evaluator.Check`` and is safe to call from any number of threads; a
dispatcher thread collects what arrives within ``max_wait_s`` of the first waiting call (or until ``max_inputs``),
evaluates it as ONE device batch and hands every caller its own outputs, in its own input order.

Calls are only batched together when their evaluation parameters are equal.  ``now`` is frozen per device batch
(the reference freezes it per call, ``evaluator_trace_common.go:24-26``): every call of a batch sees the same
instant, taken while all of them were waiting.
"""
from __future__ import annotations

import threading
import time

from .engine import DeviceUnsupported


class _Call:
    __slots__ = ("inputs", "key", "done", "outputs", "bad", "error")

    def __init__(self, inputs, key):
        self.inputs, self.key = inputs, key
        self.done = threading.Event()
        self.outputs = self.bad = self.error = None


class BatchingEvaluator:
    def __init__(self, evaluator, max_inputs: int = 4096, max_wait_s: float = 200e-6):
        self.ev = evaluator
        self.max_inputs = max_inputs
        self.max_wait_s = max_wait_s
        self._cv = threading.Condition()
        self._queue = []
        self._closed = False
        self.batches = 0          # device batches evaluated
        self.calls = 0            # calls served
        self._thread = threading.Thread(target=self._run, name="cbh-batcher", daemon=True)
        self._thread.start()

    # -- Evaluator.Check --------------------------------------------------------------------------
    def check(self, inputs, now_ns=None, lenient_scope_search=None, strict_evaluation=None,
              default_policy_version=None, default_scope=None, allow_unsupported=False):
        inputs = list(inputs)
        if not inputs:
            return ([], []) if allow_unsupported else []
        call = _Call(inputs, (now_ns, lenient_scope_search, strict_evaluation, default_policy_version, default_scope))
        with self._cv:
            if self._closed:
                raise RuntimeError("BatchingEvaluator is closed")
            self._queue.append(call)
            self._cv.notify_all()
        call.done.wait()
        if call.error is not None:
            raise call.error
        if allow_unsupported:
            return call.outputs, call.bad
        if call.bad:
            raise DeviceUnsupported(call.bad, getattr(self.ev.lt, "unsupported", []))
        return call.outputs

    def close(self):
        with self._cv:
            self._closed = True
            self._cv.notify_all()
        self._thread.join()

    # -- dispatcher -------------------------------------------------------------------------------
    def _take(self):
        """Blocks for the first call, lingers for more, returns the calls of one batch (equal parameters)."""
        with self._cv:
            while not self._queue and not self._closed:
                self._cv.wait()
            if not self._queue:
                return None
            deadline = time.monotonic() + self.max_wait_s
            while not self._closed and sum(len(c.inputs) for c in self._queue) < self.max_inputs:
                left = deadline - time.monotonic()
                if left <= 0:
                    break
                self._cv.wait(left)
            key = self._queue[0].key
            batch, rest, n = [], [], 0
            for c in self._queue:
                if c.key == key and (not batch or n + len(c.inputs) <= self.max_inputs):
                    batch.append(c)
                    n += len(c.inputs)
                else:
                    rest.append(c)
            self._queue = rest
            return batch

    def _run(self):
        while True:
            batch = self._take()
            if batch is None:
                return
            now_ns, lenient, strict, dver, dscope = batch[0].key
            flat = [i for c in batch for i in c.inputs]
            try:
                outs, bad = self.ev.check(flat, now_ns=now_ns, lenient_scope_search=lenient, strict_evaluation=strict,
                                          default_policy_version=dver, default_scope=dscope, allow_unsupported=True)
                bad = set(bad)
                at = 0
                for c in batch:
                    n = len(c.inputs)
                    c.outputs = outs[at:at + n]
                    c.bad = [i - at for i in range(at, at + n) if i in bad]
                    at += n
            except Exception as e:  # noqa: BLE001 - every waiting caller gets the failure, none is left hanging
                for c in batch:
                    c.error = e
            self.batches += 1
            self.calls += len(batch)
            for c in batch:
                c.done.set()


class _RequestCall:
    __slots__ = ("request", "aux", "key", "done", "outputs", "flags", "include_meta", "error", "effective_policies")

    def __init__(self, request, aux, key):
        self.request, self.aux, self.key = request, aux, key
        self.done = threading.Event()
        self.outputs = self.flags = self.include_meta = self.error = self.effective_policies = None


class RequestBatcher:
    """The same gathering for what a server's handler holds: the BYTES of a ``CheckResourcesRequest`` (and the serialized engine
    ``AuxData`` it derived from the request's JWT).  ``check_request`` blocks like ``svc.CheckResources``' call of ``eng.Check`` and is safe
    from any number of threads; the dispatcher sends what arrived within ``max_wait_s`` (or ``max_requests``) down the device road as ONE
    call of ``cbh_wire_check_requests_pb`` (``HipEvaluator.check_requests_pb``: the requests are split into their ``CheckInput``s on the
    device) and hands every caller the serialized ``CheckOutput``s of its own resource entries, their flags (``CBI_OUT_*``) and whether the
    request asked for ``include_meta``.  A batch the device road leaves to the host flattener is answered request by request through
    ``on_host`` (``HipEvaluator.check_request_pb`` bound by the caller), when given; else its callers get the error.
    ``audit_trail`` (a server with decision logs on): the call is ``cbh_wire_check_requests_trail_pb`` and ``check_request`` returns a
    fourth value, the keys of the request's own ``AuditTrail.EffectivePolicies`` - the coalesced batch keeps a trail per request."""

    def __init__(self, evaluator, max_requests: int = 2048, max_wait_s: float = 200e-6, on_host=None, audit_trail: bool = False):
        self.ev, self.max_requests, self.max_wait_s, self.on_host = evaluator, max_requests, max_wait_s, on_host
        self.audit_trail = audit_trail
        self._cv = threading.Condition()
        self._queue = []
        self._closed = False
        self.batches = self.calls = 0
        self._thread = threading.Thread(target=self._run, name="cbh-request-batcher", daemon=True)
        self._thread.start()

    def check_request(self, request: bytes, aux_data: bytes = None, now_ns=None, lenient_scope_search=None, strict_evaluation=None,
                      default_policy_version=None, default_scope=None):
        call = _RequestCall(bytes(request), aux_data, (now_ns, lenient_scope_search, strict_evaluation, default_policy_version, default_scope))
        with self._cv:
            if self._closed:
                raise RuntimeError("RequestBatcher is closed")
            self._queue.append(call)
            self._cv.notify_all()
        call.done.wait()
        if call.error is not None:
            raise call.error
        if self.audit_trail:
            return call.outputs, call.flags, call.include_meta, call.effective_policies
        return call.outputs, call.flags, call.include_meta

    def close(self):
        with self._cv:
            self._closed = True
            self._cv.notify_all()
        self._thread.join()

    def _take(self):
        with self._cv:
            while not self._queue and not self._closed:
                self._cv.wait()
            if not self._queue:
                return None
            deadline = time.monotonic() + self.max_wait_s
            while not self._closed and len(self._queue) < self.max_requests:
                left = deadline - time.monotonic()
                if left <= 0:
                    break
                self._cv.wait(left)
            key = self._queue[0].key
            batch = [c for c in self._queue if c.key == key][:self.max_requests]
            taken = set(map(id, batch))
            self._queue = [c for c in self._queue if id(c) not in taken]
            return batch

    def _run(self):
        from . import capi
        while True:
            batch = self._take()
            if batch is None:
                return
            now_ns, lenient, strict, dver, dscope = batch[0].key
            kw = dict(now_ns=now_ns, lenient_scope_search=lenient, strict_evaluation=strict, default_policy_version=dver, default_scope=dscope)
            try:
                aux = [c.aux for c in batch]
                got = self.ev.check_requests_pb([c.request for c in batch], aux if any(aux) else None, audit_trail=self.audit_trail, **kw)
                outs, flags, meta = got[:3]
                at = 0
                for k, (c, o, m) in enumerate(zip(batch, outs, meta)):
                    c.outputs, c.flags, c.include_meta = o, flags[at:at + len(o)], bool(m)
                    c.effective_policies = got[3][k] if self.audit_trail else None
                    at += len(o)
            except capi.HostFlattenerNeeded as e:
                for c in batch:          # (rare: an entry with more than 64 actions, a kind to rewrite that no policy names, ...)
                    if self.on_host is None:
                        c.error = e
                        continue
                    try:
                        got = self.on_host(c.request, c.aux, **kw)
                        c.outputs, c.flags, c.include_meta = got[:3]
                        c.effective_policies = got[3] if len(got) > 3 else None
                    except Exception as e2:  # noqa: BLE001
                        c.error = e2
            except Exception as e:  # noqa: BLE001 - every waiting caller gets the failure, none is left hanging
                for c in batch:
                    c.error = e
            self.batches += 1
            self.calls += len(batch)
            for c in batch:
                c.done.set()
