"""Cross-product batches: N principals x M resources x A actions decided from N + M flattened messages.

Bulk questions - "which of these users may do what to which of these documents" (audits, what-if runs over a
stored population) - are a cross product, and flattening N*M CheckInputs would spend nearly all of the time
re-parsing the same N principals and M resources (the engine is ingest bound end to end).  Here
the two halves are flattened ONCE each, in a single flattener call so that they share one batch-local string
table and one heap, and the N*M device requests are laid out with vectorised gathers: principal-side fields and
columns tile, resource-side ones repeat.  Every request carries the same actions.  The result is an ordinary
``flatten.Batch`` (``cbh_batch``), so the kernels and the resident path are unchanged.

Routing order comes for free: the kernels like requests ordered by (kind, resource version, resource scope, role
list) - here the M resources are ordered by route, the N principals by role list, and the product is laid out
resource-major: requests are then ordered by route, and by role list within every resource (all a wave needs:
its 64 lanes share the policy buckets and walk them equally often); no N*M-element sort is needed.  Device request q = j' * N + i' pairs the
j'-th resource and the i'-th principal of those orders; ``effect_cube`` maps results back to [i][j][action].
"""
from __future__ import annotations

import numpy as np

from .flatten import (RQ_ACT_CNT, RQ_ACT_OFF, RQ_KIND, RQ_NFIELDS, RQ_P_SCOPE, RQ_P_VERSION, RQ_PRINCIPAL_ID, RQ_R_SCOPE,
                      RQ_R_VERSION, RQ_ROLE_CNT, RQ_ROLE_OFF, RQ_S_KIND, RQ_S_P_SCOPE, RQ_S_P_VERSION, RQ_S_R_SCOPE,
                      RQ_S_R_VERSION, RQ_S_RESOURCE_ID, Batch)

P_FIELDS = (RQ_PRINCIPAL_ID, RQ_P_SCOPE, RQ_P_VERSION, RQ_ROLE_OFF, RQ_ROLE_CNT, RQ_S_P_SCOPE, RQ_S_P_VERSION)
R_FIELDS = (RQ_KIND, RQ_R_SCOPE, RQ_R_VERSION, RQ_S_RESOURCE_ID, RQ_S_KIND, RQ_S_R_SCOPE, RQ_S_R_VERSION)


def cross_product_batch(flattener, columns, principals, resources, actions, aux_data=None,
                        default_policy_version="default", default_scope="", sort=True) -> Batch:
    """``flattener``: ``flatten.Flattener`` or ``ingest.WireFlattener``; ``columns``: ``LoweredTable.columns``.
    ``aux_data`` (one dict, or one per principal) is visible to every request of that principal."""
    n, m, a = len(principals), len(resources), len(actions)
    if a > 64:
        raise ValueError("at most 64 actions per cross-product batch")
    aux = aux_data if isinstance(aux_data, (list, tuple)) else [aux_data] * n
    blank_p, blank_r = {"id": "", "roles": []}, {"kind": "", "id": ""}
    halves = [dict({"principal": p, "resource": blank_r, "actions": list(actions) if i == 0 else []},
                   **({"auxData": aux[i]} if aux[i] else {})) for i, p in enumerate(principals)]
    halves += [{"principal": blank_p, "resource": r, "actions": []} for r in resources]
    h = flattener.flatten(halves, default_policy_version, default_scope, sort=False)
    assert h.n_requests == n + m
    nm = n * m
    hp, hr = h.req_u32[:, :n], h.req_u32[:, n:]
    if sort and nm > 1:
        # principals by (role count, order-sensitive role-list signature), resources by (kind, version, scope):
        # the keys of flatten.sort_batch_by_route, applied to the halves
        cnt = hp[RQ_ROLE_CNT].astype(np.int64)
        sig = np.zeros(n, dtype=np.uint64)
        with np.errstate(over="ignore"):
            for i in range(n):
                o, c = int(hp[RQ_ROLE_OFF, i]), int(cnt[i])
                k = np.arange(c, dtype=np.uint64)
                sig[i] = ((h.roles[o:o + c].astype(np.uint64) + np.uint64(1)) * (k * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xC2B2AE3D27D4EB4F))).sum(dtype=np.uint64)
        p_order = np.lexsort((sig, cnt))
        r_order = np.lexsort((hr[RQ_R_SCOPE], hr[RQ_R_VERSION], hr[RQ_KIND]))
    else:
        p_order, r_order = np.arange(n), np.arange(m)
    b = Batch()
    b.n_requests, b.n_tuples, b.n_strings = nm, nm * a, h.n_strings
    req = np.zeros((RQ_NFIELDS, nm), dtype=np.uint32)
    for f in P_FIELDS:
        req[f] = np.tile(hp[f, p_order], m)
    for f in R_FIELDS:
        req[f] = np.repeat(hr[f, r_order], n)
    req[RQ_ACT_OFF] = np.arange(nm, dtype=np.uint32) * a
    req[RQ_ACT_CNT] = a
    b.req_u32 = req
    ncol = len(columns)
    b.col_tag = np.empty((ncol, nm), dtype=np.uint8)
    b.col_val = np.empty((ncol, nm), dtype=np.uint64)
    for c, (root, _) in enumerate(columns):
        if root == "R":
            b.col_tag[c], b.col_val[c] = np.repeat(h.col_tag[c, n:][r_order], n), np.repeat(h.col_val[c, n:][r_order], n)
        else:       # principal attributes and auxiliary data travel with the principal
            b.col_tag[c], b.col_val[c] = np.tile(h.col_tag[c, :n][p_order], m), np.tile(h.col_val[c, :n][p_order], m)
    act_ids = h.tuple_action[:a] if a else np.zeros(0, dtype=np.uint32)     # input 0 carried the actions
    b.tuple_action = np.tile(act_ids, nm).astype(np.uint32)
    b.tuple_req = np.repeat(np.arange(nm, dtype=np.uint32), a)
    b.roles, b.heap_tag, b.heap_val = h.roles, h.heap_tag, h.heap_val
    b.str_off, b.str_bytes, b.str_flags = h.str_off, h.str_bytes, h.str_flags
    b.vreq_input = np.arange(nm, dtype=np.int64)
    b.actions_per_request = None        # nm lists would defeat the purpose: read results through the cubes below
    b.shape = (n, m, a)
    b.p_order, b.r_order = p_order, r_order
    return b


def _cube(batch, per_tuple):
    n, m, a = batch.shape
    out = np.empty((n, m, a), dtype=per_tuple.dtype)
    out[np.ix_(batch.p_order, batch.r_order)] = np.asarray(per_tuple).reshape(m, n, a).transpose(1, 0, 2)
    return out


def effect_cube(batch, res):
    """``capi.Result`` of the batch -> uint8[n][m][a] of CBH_EFFECT_*: [i][j][k] = principals[i], resources[j], actions[k]."""
    return _cube(batch, res.effect)


def result_cubes(batch, res):
    """All per-action outputs as [n][m][a] cubes, and the derived-role masks as [n][m]."""
    n, m, _ = batch.shape
    edr = None
    if res.edr is not None:
        edr = np.empty((n, m), dtype=np.uint64)
        edr[np.ix_(batch.p_order, batch.r_order)] = np.asarray(res.edr).reshape(m, n).T
    return {f: (_cube(batch, getattr(res, f)) if getattr(res, f) is not None else None) for f in ("effect", "policy", "scope", "status")}, edr
