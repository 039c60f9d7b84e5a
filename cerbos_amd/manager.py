"""Versioned rule-table manager: the host-side mirror of ``ruletable.Manager``
(``internal/ruletable/manager.go:50-124``).

The reference keeps the current ``*RuleTable`` behind an RWMutex: ``Check`` takes the read lock for the whole
evaluation (manager.go:50-55), a storage event rebuilds the table and swaps it under the write lock
(manager.go:86-124).  Here the swap never waits for readers: a request *retains* the table it was handed
(``cbh_table_retain``), the swap publishes the new table and drops the owner's reference of the old one, and the
old image leaves HBM when its last in-flight batch has drained (tables are reference counted inside the library).
A batch is flattened against one table's string pool, so a request must use the ingest table and the device table
of the SAME version - ``acquire()`` hands out both.
"""
from __future__ import annotations

import threading

from . import capi
from .flatten import Flattener
from .lower.blob import LoweredTable


class _Version:
    """One published table: the device table (reference counted inside the library) and the host-side pieces that belong to
    it.  The ingest table is host memory only the Python side owns: it lives until the version is retired AND its last
    lease is back (a request may still be flattening / assembling against it when the swap happens)."""
    __slots__ = ("number", "lowered", "table", "flattener", "ingest", "_lock", "_leases", "_retired")

    def __init__(self, number, lowered, table, flattener, ingest):
        self.number, self.lowered, self.table, self.flattener, self.ingest = number, lowered, table, flattener, ingest
        self._lock, self._leases, self._retired = threading.Lock(), 0, False

    def _lease(self):
        with self._lock:
            self._leases += 1

    def _unlease(self):
        with self._lock:
            self._leases -= 1
            last = self._retired and self._leases == 0
        if last:
            self._close_ingest()

    def retire(self):
        """The owner's references go: the device table now (it drains in-flight batches by itself), the ingest table with
        the last lease."""
        self.table.close()
        with self._lock:
            self._retired = True
            idle = self._leases == 0
        if idle:
            self._close_ingest()

    def _close_ingest(self):
        ing, self.ingest = self.ingest, None
        if ing is not None:
            ing.close()


class TableLease:
    """One request's hold on a table version (the reference's RLock scope)."""

    def __init__(self, version: _Version):
        self.v = version
        version._lease()
        try:
            self.table = capi.Table.borrow(version.table)   # this request's own reference (cbh_table_retain)
        except BaseException:
            version._unlease()   # a lease that never came to be must not keep a retired version's ingest table alive
            raise
        self._released = False

    number = property(lambda self: self.v.number)
    lowered = property(lambda self: self.v.lowered)
    flattener = property(lambda self: self.v.flattener)
    ingest = property(lambda self: self.v.ingest)

    def release(self):
        if self._released:
            return
        self._released = True
        self.table.close()
        self.v._unlease()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()


class TableManager:
    def __init__(self, lowered: LoweredTable = None, native_ingest: bool = False):
        self._lock = threading.Lock()
        self._cur = None
        self._n = 0
        self._native_ingest = native_ingest
        if lowered is not None:
            self.swap(lowered)

    def _build(self, lowered):
        table = capi.Table(lowered.blob)
        ingest = None
        if self._native_ingest:
            from .ingest import IngestTable
            ingest = IngestTable(lowered.blob)
        return table, Flattener(lowered), ingest

    def swap(self, lowered: LoweredTable) -> int:
        """Load ``lowered`` on every device, publish it, retire the previous version (manager.go:86-124).  The
        upload and broadcast happen before the lock is taken: requests keep flowing on the old table meanwhile."""
        table, flattener, ingest = self._build(lowered)
        with self._lock:
            self._n += 1
            old, self._cur = self._cur, _Version(self._n, lowered, table, flattener, ingest)
            n = self._n
        if old is not None:
            old.retire()   # the owner's references: the image is freed once in-flight batches have drained
        return n

    def swap_pb(self, wire: bytes, globals_=None) -> int:
        """``swap`` from serialized ``runtimev1.RuleTable`` bytes - the storage-event path of the reference
        (manager.go:86-124: rebuild the table, then swap)."""
        from .lower.blob import lower_rule_table
        from .ruletable.proto import decode_rule_table
        return self.swap(lower_rule_table(decode_rule_table(wire), globals_))

    def swap_policies(self, policies: dict, sources: dict | None = None, globals_=None) -> int:
        """``swap`` from policy documents ({fqn: policy}, ``policy.loader``) - a storage event as the reference handles it
        (manager.go:86-124): compile, build the rows, lower, then swap.  A set that does not compile raises
        ``policy.compile.CompileError`` with every error and the published version stays
        (ruletable_test.go:109-140 maintain_valid_state_on_missing_derived_role)."""
        from .lower.blob import lower_rule_table
        from .ruletable.build import rule_table_from_policies
        return self.swap(lower_rule_table(rule_table_from_policies(policies, sources, require_ancestors=True), globals_))

    def acquire(self) -> TableLease:
        with self._lock:
            if self._cur is None:
                raise RuntimeError("no rule table loaded")
            return TableLease(self._cur)

    @property
    def version(self) -> int:
        with self._lock:
            return self._n

    def check_batch(self, inputs, now_ns=0, flags=capi.F_WANT_DERIVED_ROLES, default_policy_version="default", default_scope=""):
        """Flatten + decide ``inputs`` on whichever version is current when the call starts.
        Returns (version number, lowered table, batch, result)."""
        with self.acquire() as lease:
            batch = lease.flattener.flatten(inputs, default_policy_version, default_scope)
            res = lease.table.check(batch, now_ns=now_ns, flags=flags)
            return lease.number, lease.lowered, batch, res

    def close(self):
        with self._lock:
            old, self._cur = self._cur, None
        if old is not None:
            old.retire()
