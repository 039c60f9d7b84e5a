"""Rule table -> flat, column-oriented device image ("lowering", done once per table version).

Input: the rule-table dict of ``cerbos_amd.ruletable.build`` (mirror of
``runtimev1.RuleTable``).  Output: ``LoweredTable`` = the blob bytes consumed by
``cbh_table_load`` (layout: cerbos_amd/csrc/cbh_blob.h) + the host-side dictionaries the
request flattener and the response assembler need (string ids, scope indices, column
schema, policy keys, derived-role names).

What the reference computes per query with bitmap ANDs (``index/index.go:214-336``) is
precomputed here into a directory keyed by (policy kind, version, kind|principal|role,
scope): every string is an integer id, rows of one bucket are contiguous and in binding
order, glob keys become bit positions of a per-dimension automaton (``lower/globs.py``).
"""
from __future__ import annotations

import struct

import numpy as np

from .. import namer
from ..cel import parser as celparser
from ..ruletable.build import KIND_RESOURCE
from . import celc
from .celc import COND_LEAF, COND_LEAFTREE, COND_PC_MASK, LoweringError, Params, ProgramBuilder
from .globs import GlobError, GlobNFA, fix_glob, has_meta, parse_glob

BLOB_MAGIC = 0x31484243
BLOB_VERSION = 22
NONE = 0xFFFFFFFF
PAT_GLOB = 0x80000000
PAT_ANY = 0x7FFFFFFF     # the lone "*": matches every string, no automaton needed
ROW_F_ACTION_LIST, ROW_F_ROLE_LIST, ROW_F_ROLE_BY_CLASS, ROW_F_ACTION_BY_CLASS = 4, 8, 16, 32
# CbhRowField (cbh_blob.h): the hot half, then the pattern half of a rule record
(ROW_FLAGS, ROW_COND, ROW_DRCOND, ROW_POLICY, ROW_ROLE_CLASSES, _, ROW_ACTION_CLASSES, _) = range(8)
ROW_LEAF = 8           # dwords 8..15: the embedded fused-leaf record
(PAT_ACTION, PAT_ROLE, PAT_RESOURCE, PAT_COUNTS, PAT_A1, _, PAT_R1, _) = range(8)   # CbhRowPatField
ROW_F_LEAF_EMBEDDED = 64
SEC_ACTION_CLASS, SEC_ROWPAT, SEC_ROWLEAF2, SEC_DRX, SEC_REGEX = 28, 29, 30, 31, 33
SEC_TRACE_ROWS, SEC_TRACE_DR, SEC_TRACE_RP, SEC_TRACE_POOL, SEC_TRACE_STRINGS, SEC_TRACE_HOST = 34, 35, 36, 37, 38, 39
RP_F_OUTPUT_ONLY, RP_F_SHARES_KEY = 0x80000000, 0x40000000   # cbh_blob.h CBH_RP_F_*: flags in CBH_RP_ALLOW_CNT
ROW_F_DRLEAF_EMBEDDED = 128
ROW_F_TREE_EMBEDDED, ROW_F_DRTREE_EMBEDDED = 256, 512   # the slot holds a tree descriptor (_tree_descriptor)
MF_FLAT_CLOSED = 512
# cbh_check_walk2.h: the decision kernel for everything else a table can hold (principal policies, role policies, parent
# roles, glob patterns, generic programs).  Per record: CBH_SEC_ROWX; per role-policy rule: CBH_SEC_RPX.
MF_WALK2 = 2048
ROW_F_X, ROW_F_XEXACT, ROW_F_OUTPUT = 1024, 2048, 4096
MF_TRACE_ALL = 4096    # the older kernels cannot tell which inputs of this table need the trace pass: they mark every one
SEC_ROWX, SEC_RPX, SEC_STR_WFLAGS = 40, 41, 42
B_RPROLES, B_FAMILY = 8, 9
B_RESSEG = 10                         # (ver sid, kind sid, scope idx) -> v0 first segment block (16-dword units of CBH_SEC_SEGS), v1 segments, v2 dr_begin, v3 dr_count
SEC_SEGS, SEC_LEAFPOOL = 43, 44
M_SEGS = 23                           # CBH_M_SEGS: M_SEGS_PRESENT | M_SEGS_POOLED | leaves in CBH_SEC_LEAFPOOL
M_SEGS_PRESENT, M_SEGS_POOLED, M_SEGS_ITEMS_POOLED = 1 << 31, 1 << 30, 1 << 29
SEG_RECORDS = 64                      # records (and distinct conditions, and distinct leaves) per segment: one bit each of a 64-bit mask
SEG_TAIL_PAD = 20                     # 16-dword units behind the last block: every lane loads an item descriptor slot, whatever n_items
SEG_COMPLEX = 16                      # items of a segment the lanes cannot decide by themselves (deeper trees, more than four leaves, programs)
SEG_FIXED_DWORDS = 176                # header 16 + class masks 128 + the two record -> item tables 32
SWF_PRINCIPAL, SWF_PARENTS = 1, 2     # CBH_SEC_STR_WFLAGS bits
SCOPE_F_ROLEPOL = 16                  # CBH_SEC_SCOPE_FLAGS bit 4: some role policy lives at this scope
M_GSLOTS_GENERIC, M_GSLOTS_ALL, M_INLINE_COLS, M_SENS_COLS, M_Q_SITES = 18, 19, 20, 21, 22
GSLOT_NONE = 0xFFFF
WALK2_MAX_GLOBS = 16      # glob patterns per dimension (action, role) a lane keeps match bits for
WALK2_MAX_GSLOTS = 256    # evaluation-site slots of one request (4 result bits each, 16 to a 64-bit word: cbh_walk2_pre_kernel)

(SEC_META, SEC_STR_OFF, SEC_STR_BYTES, SEC_SCOPE_PARENT, SEC_SCOPE_FLAGS, SEC_SCOPE_SID, SEC_HASH,
 SEC_ROWS, SEC_RPROWS, SEC_U32POOL, SEC_DR, SEC_CODE, SEC_CONST_TAG, SEC_CONST_VAL, SEC_THEAP_TAG,
 SEC_THEAP_VAL, SEC_GBITS, SEC_NFA_ACTION, SEC_NFA_ROLE, SEC_NFA_KIND, SEC_POLICY_SID,
 SEC_DRNAME_SID, SEC_CONST_REC, SEC_THEAP_REC, SEC_ROLE_CLASS, SEC_COLUMN_PATHS, SEC_HOST_NAMES) = range(1, 28)

(M_NSTRINGS, M_NCOLUMNS, M_NSCOPES, M_HASH_MASK, M_NROWS, M_NRPROWS, M_NDR, M_NPOLICIES, M_NCONSTS,
 M_CODE_LEN, M_FLAGS, M_MAX_STACK, M_NDRNAMES, M_NFA_WORDS_ACTION, M_NFA_WORDS_ROLE,
 M_NFA_WORDS_KIND, M_THEAP_LEN, M_MAX_LOCALS) = range(18)
META_N = 24
MF_USES_RUNTIME_EDR = 1

B_RESOURCE, B_PRINCIPAL, B_ROLEPOL, B_PPEXISTS, B_RPRES, B_PARENTS, B_RESEXISTS = range(1, 8)

DIM_ACTION, DIM_ROLE, DIM_KIND = 0, 1, 2
FLAG_RES, FLAG_PRIN = 1, 2


def hash4(a, b, c, d):
    """Must match hash4() in cbh_engine.hip."""
    M = 0xFFFFFFFF
    h = (a * 0x9E3779B1) & M
    h = ((h ^ (h >> 15)) + b * 0x85EBCA77) & M
    h = ((h ^ (h >> 13)) + c * 0xC2B2AE3D) & M
    h = ((h ^ (h >> 16)) + d * 0x27D4EB2F) & M
    h ^= h >> 15
    h = (h * 0x2C1B3C6D) & M
    h ^= h >> 12
    return h


class _Dim:
    """Pattern set of one index dimension (action / role / resource kind).  The device keeps ONE 64-bit word of match bits per
    string and dimension and an automaton of at most eight state words: a pattern beyond either (`overflow`) gets no index."""

    def __init__(self):
        self.globs = {}  # pattern text -> glob index
        self.positions = 0
        self.overflow = set()

    def glob_ref(self, pattern, may_overflow=False):
        gi = self.globs.get(pattern)
        if gi is None:
            if pattern in self.overflow:
                return None
            try:
                cost = sum(len(seq) + 1 for seq in parse_glob(fix_glob(pattern)))
            except GlobError:
                cost = 0     # an invalid glob never matches (globs_common.go:33-36): it takes an index, no states
            if may_overflow and (len(self.globs) >= GlobNFA.MAX_GLOBS or self.positions + cost > 64 * GlobNFA.MAX_WORDS):
                self.overflow.add(pattern)
                return None
            gi = len(self.globs)
            self.globs[pattern] = gi
            self.positions += cost
        return PAT_GLOB | gi


class LoweredTable:
    def __init__(self):
        self.blob = b""
        self.strings = []       # id -> str
        self.string_ids = {}    # str -> id
        self.scopes = []        # scope index -> scope string
        self.scope_index = {}
        self.columns = []       # column index -> (root, keys)
        self.policy_keys = []   # policy id -> policy key string
        self.dr_names = []      # bit -> derived role name
        self.unsupported = []   # [(expression, reason)] compiled to OP_UNSUPPORTED
        self.nfas = [None, None, None]
        self.per_call_globals = False   # `G.x` read from the call's globals (a column of root "G"), not folded into the image
        self.stats = {}

    def sid(self, s):
        return self.string_ids.get(s)


def _cond_uses_runtime(cond, params: Params):
    if cond is None:
        return False
    if cond[0] == "expr":
        ast = params.inline(celparser.parse(cond[1]))
        return any(n[0] == "ident" and n[1] == "runtime" for n in celparser.walk(ast))
    return any(_cond_uses_runtime(c, params) for c in cond[1])


def lower_rule_table(rt: dict, globals_=None, trace=True, per_call_globals=False) -> LoweredTable:
    """``per_call_globals``: `G.x` is not folded into the image but read from the globals every CALL brings
    (``Flattener.flatten(globals_=...)``, ``cbi_flatten_pb_g``, ``cbh_wire_flatten``'s ``globals_pb``): one image, any globals.

    The columns the kernels' inline leaf code reads get the lowest indices: the walk that runs no generic program
    (cbh_walk2_kernel) parks only those in LDS, and LDS is what bounds its occupancy.  Which columns those are is known once
    the programs are compiled - hence up to three passes, the later ones with the column order of the one before."""
    first = ()
    if per_call_globals:
        from .celc import PER_CALL_GLOBALS
        globals_ = PER_CALL_GLOBALS
    for _ in range(3):
        lt = _lower_rule_table(rt, globals_, trace, first)
        want = tuple(lt.columns[i] for i in sorted(lt.inline_cols))
        if want == tuple(lt.columns[:len(want)]):
            break
        first = want
    return lt


def _lower_rule_table(rt: dict, globals_, trace, first_columns) -> LoweredTable:  # noqa: C901
    lt = LoweredTable()
    from .celc import _PerCallGlobals
    globals_ = globals_ if isinstance(globals_, _PerCallGlobals) else dict(globals_ or {})

    def sid(s):
        i = lt.string_ids.get(s)
        if i is None:
            i = len(lt.strings)
            lt.string_ids[s] = i
            lt.strings.append(s)
        return i

    sid("")  # string 0 is always the empty string
    pb = ProgramBuilder(sid, globals_)
    for root, keys in first_columns:
        pb.column(root, keys)
    dims = [_Dim(), _Dim(), _Dim()]
    used_any = []   # dimensions in which a lone "*" was met

    def dim_ref(dim, key):
        """globDimension.Set: a key is a pattern only if it contains '*' (glob_dimension.go:32)."""
        if key == "*":
            used_any.append(dim)
            return PAT_ANY          # fixGlob: "*" means "**" (util/globs_common.go:74-81)
        if "*" in key:
            ref = dims[dim].glob_ref(key, may_overflow=dim != DIM_KIND)
            if ref is None:           # beyond what the device holds for the dimension: see glob_overflows
                used_any.append(dim)
                return PAT_ANY
            return ref
        return sid(key)

    def allow_action_ref(a):
        """Role-policy allow actions are matched with ``a == action || MatchesGlob(a, action)``
        (index.go:447-452): any glob metacharacter makes it a pattern."""
        if has_meta(a):
            ref = dims[DIM_ACTION].glob_ref(a, may_overflow=True)
            if ref is None:
                used_any.append(DIM_ACTION)
                return PAT_ANY
            return ref
        return sid(a)

    def glob_overflows(dim, keys, meta_is_glob=False):
        """Does a role / action list hold a pattern the dimension has no room for (more than 64 patterns, or an automaton beyond
        512 positions)?  Such a pattern is lowered as "*" - it matches MORE than it should - and the rule's condition as a
        program that flags whoever evaluates it (CBH_ST_UNSUPPORTED): every request that could have matched the real pattern
        is handed back to the caller's own engine, never answered wrongly, and the rest of the table still serves (the
        reference has no such limit: index/glob_dimension.go:31-118)."""
        hit = False
        for k in keys or ():
            if k and k != "*" and (("*" in k) or (meta_is_glob and has_meta(k))):
                if dims[dim].glob_ref(k, may_overflow=True) is None:
                    hit = True
        return hit

    # ---- scopes: every scope of a row or role policy plus all ancestors; "" is index 0
    scope_set = {""}
    for r in rt["rules"]:
        scope_set.add(r["scope"])
    for s in rt["scope_parent_roles"]:
        scope_set.add(s)
    for s in list(scope_set):
        scope_set.update(namer.scope_parents(s))
    lt.scopes = sorted(scope_set, key=lambda s: (s.count(".") if s else -1, s))
    lt.scope_index = {s: i for i, s in enumerate(lt.scopes)}
    res_scopes, prin_scopes = set(rt["resource_scopes"]), set(rt["principal_scopes"])
    scope_parent, scope_flags, scope_sid = [], [], []
    for s in lt.scopes:
        if s == "":
            scope_parent.append(NONE)
        else:
            scope_parent.append(lt.scope_index[next(iter(namer.scope_parents(s)))])
        fl = (FLAG_RES if s in res_scopes else 0) | (FLAG_PRIN if s in prin_scopes else 0)
        fl |= (rt["scope_permissions"].get(s, 0) & 3) << 2
        scope_flags.append(fl)
        scope_sid.append(sid(s))

    # ---- policies (ids used for role-policy / strict-mode attribution)
    policy_index = {}

    def policy_id(fqn):
        key = namer.policy_key_from_fqn(fqn)
        i = policy_index.get(key)
        if i is None:
            i = len(lt.policy_keys)
            policy_index[key] = i
            lt.policy_keys.append(key)
        return i

    # ---- bucket the rows
    res_buckets, prin_buckets, rp_buckets = {}, {}, {}
    res_exists, pp_exists, rp_res = set(), set(), {}
    rp_evalkeys = {}
    rp_history_dependent = set()   # ids of role-policy rules whose cached condition outcome depends on evaluation history
    rp_shares_key = set()          # ids of role-policy rules sharing their key across "conditional" / "output only"
    for r in rt["rules"]:
        ver, scope = r["version"], r["scope"]
        if r["allow_actions"] is not None:
            rp_buckets.setdefault((ver, scope, r["role"]), []).append(r)
            rp_res.setdefault((ver, scope), []).append(r["resource"])
            # Only conditional rules are ever evaluated (and cached) through their key (index.go:463-487).  The key
            # leaves the resource out (ruletable.go:445-455), so rules of one role policy for different resources
            # share it; the per-request conditionCache (check.go:186, 324) can only confuse them when both rules
            # match the SAME resource kind, i.e. when one of the two resource names is a glob.
            # Such rules are not refused wholesale: their conditions become UNSUPPORTED programs (the requests that
            # reach them are flagged for the caller's own engine), the rest of the table serves as usual.
            # (A rule without a condition but with an output expression is visited as well - as a binding without effect,
            # index.go:463-484 - and caches "satisfied" under the shared key: it takes part like a condition that is None.)
            # Against a conditional rule that is one more way to confuse the cache, and one the device can follow: whichever
            # of the two the request's actions reach first decides what both see (cbh_check_wave.h CBH_RP_F_*).
            if r["condition"] is not None or r["emit_output"]:
                for o in rp_evalkeys.setdefault(r["evaluation_key"], []):
                    if o["condition"] != r["condition"] and ("*" in o["resource"] or "*" in r["resource"]):
                        if o["condition"] is not None and r["condition"] is not None:
                            rp_history_dependent.add(o["id"]); rp_history_dependent.add(r["id"])
                        else:
                            rp_shares_key.add(o["id"]); rp_shares_key.add(r["id"])
                rp_evalkeys[r["evaluation_key"]].append(r)
        elif r["policy_kind"] == KIND_RESOURCE:
            if "*" in r["resource"]:
                raise LoweringError("resource policy with a wildcard resource name is not supported: %s" % r["resource"])
            key = (ver, r["resource"], scope)
            res_exists.add(key)
            res_buckets.setdefault(key, [])
            if r["action"] is not None:
                res_buckets[key].append(r)
        else:
            pp_exists.add((ver, scope))
            if r["action"] is not None:
                prin_buckets.setdefault((ver, scope, r["principal"]), []).append(r)

    pool = []
    row_cols = [[] for _ in range(16)]
    pat_cols = [[] for _ in range(8)]
    row_roles = []   # per device row: its role strings (None for principal-policy rows)
    row_actions = []  # likewise its action strings
    rp_cols = [[] for _ in range(4)]
    dr_cols = [[] for _ in range(4)]
    dr_parents = []   # per derived-role record: its parent role strings
    entries = []  # (k0,k1,k2,k3, v0,v1,v2,v3)
    trace_row_rules = []   # per device row: (a rule-table row of its rule, principal policy?) - the trace pass's programs are compiled last
    trace_dr_defs = []     # per derived-role record: its definition
    trace_rp_rules = []    # per role-policy record: its rule
    row_family = []        # per device row: ("R", version, kind) | ("P", version, principal) - whose evaluation sites share slot numbers
    row_scope = []         # per device row: the scope of its policy
    rp_scope = []          # per role-policy record likewise
    rp_family = []         # per role-policy record: its version
    rp_allow = []          # per role-policy record: its allow-action strings
    dr_family = []         # per derived-role record: ("R", version, kind)
    row_probe = []         # per device row: (probe of its params' variables, of its derived-role params' variables) - program or None
    dr_probe = []          # per derived-role record: probe of the definition's variables
    rp_bucket_probe = {}   # directory entry index of a role-policy bucket -> probe of the policy's variables

    def row_programs(r, principal_policy):
        params = Params(r["params"]["constants"], r["params"]["ordered_variables"], globals_) if r["params"] else Params(None, None, globals_)
        if glob_overflows(DIM_ROLE, [r["role"]]) | glob_overflows(DIM_ACTION, [r["action"]]):
            cond = pb.unsupported_program(namer.policy_key_from_fqn(r["origin_fqn"]),
                                          "more glob patterns in the role / action dimension than the device table holds (64 patterns, 512 automaton positions)")
        elif principal_policy and _cond_uses_runtime(r["condition"], params):
            # in the reference the value then depends on the previously evaluated action (check.go:281), which a
            # per-tuple evaluation cannot reproduce: whoever reaches this rule is flagged, not answered
            cond = pb.unsupported_program(namer.policy_key_from_fqn(r["origin_fqn"]),
                                          "principal policy condition reads runtime.effectiveDerivedRoles (history dependent, check.go:281)")
        else:
            cond = pb.condition_program(r["condition"], params) if r["condition"] is not None else NONE
        drc = NONE
        if r["derived_role_condition"] is not None:
            dp = r["derived_role_params"] or {"constants": {}, "ordered_variables": []}
            drc = pb.condition_program(r["derived_role_condition"], Params(dp["constants"], dp["ordered_variables"], globals_))
        return cond, drc

    def dim_list(dim, keys):
        """-> (first word, count, list flag, [2nd, 3rd inline refs]).  One key: its reference, count 0.
        Up to three: all inline in the record.  More: a slice of the u32 pool."""
        if len(keys) == 1:
            return (dim_ref(dim, keys[0]) if keys[0] else NONE), 0, False, [NONE] * 2
        refs = [dim_ref(dim, k) for k in keys]
        if len(refs) <= 3:
            return refs[0], len(refs), False, (refs[1:] + [NONE] * 2)[:2]
        off = len(pool)
        pool.extend(refs)
        return off, len(refs), True, [NONE] * 2

    def add_bucket_rows(rows, principal_policy, family=None, scope=None):
        """Emit the device rows of one bucket; returns how many.

        The rule table holds one row per (rule, role, action) (ruletable.go addResourcePolicy /
        addPrincipalPolicy).  Rows of ONE rule that share effect, condition and derived-role condition
        differ only in (role, action) and form the cross product roles x actions, so the device walks
        them as a single record carrying both lists: the condition is evaluated once per request instead
        of once per (role, action) row.  A rule's rows are contiguous and each of its signature groups
        keeps its first-occurrence position, so for every (role, action) the matching records are met
        in the order Index.Query returns the rows (index/index.go:214-336)."""
        blocks = []   # [(rule identity, {signature: [rows]})] in binding order
        for r in rows:
            ident = (r["origin_fqn"], r["name"])
            if not blocks or blocks[-1][0] != ident:
                blocks.append((ident, {}))
            cond, drc = row_programs(r, principal_policy)
            # a rule with an output expression keeps the reference's rows one by one: every visit of a row emits an
            # OutputEntry (check.go:383-411), so their number and order are observable
            sig = (r["resource"], r["effect"], cond, drc) + ((r["id"],) if trace and r["emit_output"] else ())
            blocks[-1][1].setdefault(sig, []).append(r)
        n = 0
        for _ident, groups in blocks:
            for (resource, effect, cond, drc, *_row), grp in groups.items():
                roles = list(dict.fromkeys(r["role"] for r in grp))
                actions = list(dict.fromkeys(r["action"] for r in grp))
                pairs = {(r["role"], r["action"]) for r in grp}
                if len(pairs) == len(roles) * len(actions) and len(actions) < 0x10000 and len(roles) < 0x10000:
                    merged = [(roles, actions)]
                else:   # not a cross product: keep the reference's rows one by one
                    merged = [([r["role"]], [r["action"]]) for r in grp]
                for rl, al in merged:
                    a_ref, a_cnt, a_pool, a_more = dim_list(DIM_ACTION, al)
                    r_ref, r_cnt, r_pool, r_more = dim_list(DIM_ROLE, rl)
                    fl = {"ALLOW": 1, "DENY": 2}.get(effect, 0)
                    fl |= (ROW_F_ACTION_LIST if a_pool else 0) | (ROW_F_ROLE_LIST if r_pool else 0)
                    row_cols[ROW_FLAGS].append(fl)
                    row_cols[ROW_COND].append(cond)
                    row_cols[ROW_DRCOND].append(drc)
                    row_cols[ROW_POLICY].append(policy_id(grp[0]["origin_fqn"]))
                    pat_cols[PAT_ACTION].append(a_ref)
                    pat_cols[PAT_ROLE].append(r_ref)
                    pat_cols[PAT_RESOURCE].append(dim_ref(DIM_KIND, resource) if resource else NONE)
                    pat_cols[PAT_COUNTS].append(a_cnt | (r_cnt << 16))
                    for i in range(2):
                        pat_cols[PAT_A1 + i].append(a_more[i])
                        pat_cols[PAT_R1 + i].append(r_more[i])
                    row_roles.append(None if principal_policy else rl)
                    row_actions.append(None if principal_policy else al)
                    trace_row_rules.append((grp[0], principal_policy))
                    row_family.append(family)
                    row_scope.append(scope)
                    r0 = grp[0]
                    pa = pb.vars_probe_program(Params(r0["params"]["constants"], r0["params"]["ordered_variables"], globals_)) if r0["params"] else None
                    pd = None
                    if r0["derived_role_condition"] is not None and r0["derived_role_params"]:
                        dp0 = r0["derived_role_params"]
                        pd = pb.vars_probe_program(Params(dp0["constants"], dp0["ordered_variables"], globals_))
                    row_probe.append((pa, pd))
                    if r0["emit_output"]:
                        row_cols[ROW_FLAGS][-1] |= ROW_F_OUTPUT
                    n += 1
        return n

    # resource policies (+ their derived roles)
    for key in sorted(res_buckets):
        ver, kind, scope = key
        rows = sorted(res_buckets[key], key=lambda r: r["id"])
        begin = len(row_cols[0])
        n_rows = add_bucket_rows(rows, False, ("R", ver, kind), scope)
        dr_begin = len(dr_cols[0])
        drs = rt["policy_derived_roles"].get(namer.resource_policy_fqn(kind, ver, scope)) or {}
        for name, dr in drs.items():
            bit = pb.dr_bit(name)
            if bit is None:
                # A 65th derived role: the mask has no bit to report it with.  Its definition stays, under bit 63, with a condition
                # that flags whoever evaluates it - the requests whose roles reach the definition go back to the caller's own
                # engine (their effectiveDerivedRoles could not be told), every other request of the table is answered.
                dr = dict(dr, condition=None)
            dr_cols[0].append(63 if bit is None else bit)
            dr_parents.append(list(dr["parent_roles"]))
            if "*" in dr["parent_roles"]:
                dr_cols[1].append(0)
                dr_cols[2].append(NONE)
            else:
                dr_cols[1].append(len(pool))
                dr_cols[2].append(len(dr["parent_roles"]))
                pool.extend(sid(p) for p in dr["parent_roles"])
            trace_dr_defs.append(dr)
            dr_family.append(("R", ver, kind))
            dr_probe.append(pb.vars_probe_program(Params(dr["constants"], dr["ordered_variables"], globals_, null_on_error=True)))
            dparams = Params(dr["constants"], dr["ordered_variables"], globals_, null_on_error=True)
            # a derived-role definition that reads runtime.effectiveDerivedRoles sees, in the reference, the roles of the scope
            # visited just before (check.go:237-282: evalCtx is replaced AFTER a scope's definitions were evaluated, a scope is
            # evaluated once per request, and every walk visits the chain in order) - the deepest scope sees none.  That is
            # what Lane.edr holds while the general walk evaluates a scope's definitions (cbh_check_wave.h), and tables whose
            # programs read runtime.* stay on that kernel.
            if bit is None:
                dr_cols[3].append(pb.unsupported_program(namer.resource_policy_fqn(kind, ver, scope) + "#" + name,
                                                         "more than 64 distinct derived role names: no bit of the effective-derived-roles mask is left for this one"))
            else:
                dr_cols[3].append(pb.condition_program(dr["condition"], dparams, allow_runtime=True)
                                  if dr["condition"] is not None else NONE)
        entries.append((B_RESOURCE, sid(ver), sid(kind), lt.scope_index[scope],
                        begin, n_rows, dr_begin, len(dr_cols[0]) - dr_begin))
    for ver, kind, scope in sorted(res_exists):
        entries.append((B_RESEXISTS, sid(ver), sid(kind), lt.scope_index[scope], 1, 0, 0, 0))

    # A request whose principal id is EMPTY (protovalidate rejects it on the gRPC path; an in-process caller of engine.Check can
    # still bring one): Index.Query leaves the principal dimension out of the AND when the id is "" (index/index.go:228-234),
    # so the rows of EVERY principal policy of (version, scope) answer it, in binding order.  The directory is keyed by the
    # exact principal string, so that union is one more bucket under the empty string (no policy can be written for it:
    # the compiler requires a principal name).
    for (ver, scope) in sorted({(v, s) for (v, s, _p) in prin_buckets}):
        if (ver, scope, "") not in prin_buckets:
            prin_buckets[(ver, scope, "")] = [r for (v, s, _p), rows in prin_buckets.items() if (v, s) == (ver, scope) for r in rows]

    # principal policies
    for key in sorted(prin_buckets):
        ver, scope, principal = key
        rows = sorted(prin_buckets[key], key=lambda r: r["id"])
        begin = len(row_cols[0])
        n_rows = add_bucket_rows(rows, True, ("P", ver, principal), scope)
        entries.append((B_PRINCIPAL, sid(ver), lt.scope_index[scope], sid(principal), begin, n_rows, 0, 0))
    for ver, scope in sorted(pp_exists):
        entries.append((B_PPEXISTS, sid(ver), lt.scope_index[scope], 0, 1, 0, 0, 0))

    # role policies
    for key in sorted(rp_buckets):
        ver, scope, role = key
        rows = sorted(rp_buckets[key], key=lambda r: r["id"])
        begin = len(rp_cols[0])
        for r in rows:
            params = Params(r["params"]["constants"], r["params"]["ordered_variables"], globals_)
            rp_cols[0].append(dim_ref(DIM_KIND, r["resource"]))
            rp_cols[1].append(len(pool))
            rp_cols[2].append(len(r["allow_actions"])
                              | (RP_F_OUTPUT_ONLY if (r["condition"] is None and r["emit_output"]) else 0)
                              | (RP_F_SHARES_KEY if r["id"] in rp_shares_key else 0))
            pool.extend(allow_action_ref(a) for a in r["allow_actions"])
            trace_rp_rules.append(r)
            rp_family.append(ver)
            rp_scope.append(scope)
            rp_allow.append(list(r["allow_actions"]))
            if glob_overflows(DIM_ACTION, r["allow_actions"], meta_is_glob=True):
                rp_cols[3].append(pb.unsupported_program(namer.policy_key_from_fqn(r["origin_fqn"]),
                                                         "more glob patterns among the allow actions than the device table holds (64 patterns, 512 automaton positions)"))
            elif r["id"] in rp_history_dependent:
                rp_cols[3].append(pb.unsupported_program(namer.policy_key_from_fqn(r["origin_fqn"]),
                                                         "role-policy rules for overlapping resource globs share an evaluation key but "
                                                         "not a condition (history dependent, ruletable.go:445-455)"))
            else:
                rp_cols[3].append(pb.condition_program(r["condition"], params) if r["condition"] is not None else NONE)
        pid = policy_id(namer.role_policy_fqn(role, ver, scope))
        r0 = rows[0]
        rpp = pb.vars_probe_program(Params(r0["params"]["constants"], r0["params"]["ordered_variables"], globals_)) if r0["params"] else None
        if rpp is not None:
            rp_bucket_probe[len(entries)] = (rpp, ver, scope)
        entries.append((B_ROLEPOL, sid(ver), lt.scope_index[scope], sid(role), begin, len(rows), pid, NONE))
    for (ver, scope), pats in sorted(rp_res.items()):
        uniq = list(dict.fromkeys(pats))
        entries.append((B_RPRES, sid(ver), lt.scope_index[scope], 0, len(pool), len(uniq), 0, 0))
        pool.extend(dim_ref(DIM_KIND, p) for p in uniq)

    # parent roles (index.go:749-788), looked up with the request's own resource scope
    for scope, roles in sorted(rt["parent_roles"].items()):
        for role, ancestors in sorted(roles.items()):
            if not ancestors:
                continue
            entries.append((B_PARENTS, lt.scope_index[scope], sid(role), 0, len(pool), len(ancestors), 0, 0))
            pool.extend(sid(a) for a in ancestors)

    # ---- the trace pass's programs (cbh_trace_batch; cbh_blob.h CBH_SEC_TRACE_*), compiled after every decision program so
    # that the columns only they read (variables nothing references, output expressions) come last
    trace_pool, trace_rows, trace_dr, trace_rp = [], [], [], []
    if trace:
        var_slices = {}
        rule_ids = {}   # (evaluation key, rule FQN) -> the small id output records carry (role-policy rules can share a key)

        def tparams(p):
            return Params(p["constants"], p["ordered_variables"], globals_, trace=True) if p else Params(None, None, globals_, trace=True)

        def var_slice(tp):
            k = tp.key()
            if k not in var_slices:
                pcs = pb.trace_variable_programs(tp)
                var_slices[k] = (len(trace_pool), len(pcs))
                trace_pool.extend(pcs)
            return var_slices[k]

        for r, principal_policy in trace_row_rules:
            tp = tparams(r["params"])
            if principal_policy and _cond_uses_runtime(r["condition"], Params(tp.constants, tp.ordered_variables, globals_)):
                cond = pb.trace_unsupported_program(namer.policy_key_from_fqn(r["origin_fqn"]), "history dependent (check.go:281)")
            else:
                cond = pb.trace_condition_program(r["condition"], tp) if r["condition"] is not None else NONE
            voff, vcnt = var_slice(tp) if r["params"] else (0, 0)
            drc, doff, dcnt = NONE, 0, 0
            if r["derived_role_condition"] is not None:
                dtp = tparams(r["derived_role_params"])
                drc = pb.trace_condition_program(r["derived_role_condition"], dtp)
                doff, dcnt = var_slice(dtp) if r["derived_role_params"] else (0, 0)
            out_act = out_not = NONE
            emit = r["emit_output"] or {}
            if emit:
                m = rt["meta"].get(r["origin_fqn"])
                src = namer.rule_fqn(m["kind"], m["name"], m["version"], r["scope"], r["name"]) if m else ""
                rule_id = rule_ids.setdefault((r["evaluation_key"], src), len(rule_ids))
                if rule_id >= 1 << 23:
                    raise LoweringError("more than 2^24 rules with outputs")
                if emit.get("rule_activated"):
                    out_act = pb.trace_output_program(emit["rule_activated"], tp, src, rule_id, False)
                if emit.get("condition_not_met"):
                    out_not = pb.trace_output_program(emit["condition_not_met"], tp, src, rule_id, True)
            trace_rows.append([cond, drc, voff, vcnt, doff, dcnt, out_act, out_not])
        for dr in trace_dr_defs:
            dtp = Params(dr["constants"], dr["ordered_variables"], globals_, trace=True, null_on_error=True)
            voff, vcnt = var_slice(dtp)
            trace_dr.append([pb.trace_condition_program(dr["condition"], dtp, allow_runtime=True) if dr["condition"] is not None else NONE,
                             voff, vcnt, 0])
        for r in trace_rp_rules:
            tp = tparams(r["params"])
            voff, vcnt = var_slice(tp) if r["params"] else (0, 0)
            if r["id"] in rp_history_dependent:
                cond = pb.trace_unsupported_program(namer.policy_key_from_fqn(r["origin_fqn"]), "history dependent (ruletable.go:445-455)")
            else:
                cond = pb.trace_condition_program(r["condition"], tp) if r["condition"] is not None else NONE
            out_act = out_not = NONE
            emit = r.get("emit_output") or {}
            if emit:
                # the synthetic row the reference evaluates is none(condition) with the two outputs swapped (index.go:436-530):
                # taken together, the rule's own ruleActivated fires when its condition holds (or it has none),
                # conditionNotMet when it does not - which is how the device walks it
                m = rt["meta"].get(r["origin_fqn"])
                src = namer.rule_fqn(m["kind"], m["name"], m["version"], r["scope"], r["name"]) if m else ""
                rule_id = rule_ids.setdefault((r["evaluation_key"], src), len(rule_ids))
                if rule_id >= 1 << 23:
                    raise LoweringError("more than 2^23 rules with outputs")
                if emit.get("rule_activated"):
                    out_act = pb.trace_output_program(emit["rule_activated"], tp, src, rule_id, False)
                if emit.get("condition_not_met"):
                    out_not = pb.trace_output_program(emit["condition_not_met"], tp, src, rule_id, True)
            trace_rp.append([cond, voff, vcnt, out_act, out_not, 0, 0, 0])
    lt.theap = (list(pb.theap_tag), [int(v) & 0xFFFFFFFFFFFFFFFF for v in pb.theap_val])   # constant lists / maps an output may yield
    lt.trace_strings = list(pb.trace_strings)
    lt.trace_templates = dict(pb.trace_templates)   # what the outputs' constructors look like (host side of trace.py)
    lt.trace_unsupported = list(dict.fromkeys(pb.trace_unsupported))
    # does the table ask for the trace pass beyond the tuples the decision kernels mark CBH_ST_CEL_ERROR?  Variables are
    # evaluated whether or not a condition reads them (check.go:651-677), outputs on every visit of their rule
    lt.trace_has_outputs = any(tr[6] != NONE or tr[7] != NONE for tr in trace_rows) or any(tr[3] != NONE or tr[4] != NONE for tr in trace_rp)
    lt.trace_has_variables = bool(trace_pool)


    # ---- glob automata + match bits of the table's own strings
    gbits = np.zeros((3, 0), dtype=np.uint64)
    nfa_bytes = []
    for d in range(3):
        pats = [p for p, _ in sorted(dims[d].globs.items(), key=lambda kv: kv[1])]
        nfa = GlobNFA(pats)
        lt.nfas[d] = nfa
        nfa_bytes.append(nfa.tables())
    K = len(lt.strings)  # final: no string may be interned after this point
    gbits = np.zeros((3, K), dtype=np.uint64)
    for d in range(3):
        if lt.nfas[d].patterns:
            for i, s in enumerate(lt.strings):
                gbits[d, i] = lt.nfas[d].match_bits(s.encode("utf-8"))

    # ---- role classes: every literal role of a rule gets a small class number, every rule record the
    # mask of the classes its role list can match.  A wave ORs the classes of the roles it is walking
    # and skips, on the scalar unit, the records none of them can match (cbh_check_wave.h).
    # Class 63 = "any other string"; a glob role or a role beyond 62 classes matches everything.
    def classes(lists, also=()):
        """class numbers for the literal strings of `lists` and `also` (first come, first numbered; at most 62), the
        u8[K] lookup table, and per row (mask of classes the list can match, does the mask decide the match exactly)."""
        class_of = {}
        for lst in list(lists) + list(also):
            for key in lst or ():
                if key and "*" not in key and key not in class_of and len(class_of) < 62:
                    class_of[key] = len(class_of)
        table = np.full(K, 63, dtype=np.uint8)
        for key, cls in class_of.items():
            table[lt.string_ids[key]] = cls
        per_row = []
        for lst in lists:
            mask, exact = 0, lst is not None
            for key in lst or [None]:
                if key == "*":                      # matches every string: every bit, still exact
                    mask = 0xFFFFFFFFFFFFFFFF
                elif not key or "*" in key:         # a glob (or no list: principal-policy row): cannot be told by class
                    mask, exact = 0xFFFFFFFFFFFFFFFF, False
                else:
                    mask |= 1 << class_of.get(key, 63)
                    exact = exact and key in class_of
            per_row.append((mask, exact))
        return table, per_row, len(class_of)

    rp_role_names = list(dict.fromkeys(role for (_v, _s, role) in sorted(rp_buckets)))
    # parent roles of derived roles, the roles that have role policies and the literal actions of their allow lists get classes too
    role_class, role_rows, n_role_classes = classes(row_roles, dr_parents + [rp_role_names])
    action_class, action_rows, n_action_classes = classes(row_actions, [[a for a in al if not has_meta(a)] for al in rp_allow])
    # fewer than 32 classes in both dimensions: "any other string" (bit 63) is mirrored in bit 31 of the low dwords,
    # so that a kernel may match on the low dword alone (cbh_check_flat.h); no lane ever holds class 31 itself
    small_classes = n_role_classes < 31 and n_action_classes < 31
    if small_classes:
        role_rows = [(m | ((m >> 63) << 31), e) for m, e in role_rows]
        action_rows = [(m | ((m >> 63) << 31), e) for m, e in action_rows]
    for i, ((rmask, rexact), (amask, aexact)) in enumerate(zip(role_rows, action_rows)):
        row_cols[ROW_ROLE_CLASSES].append(rmask & 0xFFFFFFFF)
        row_cols[ROW_ROLE_CLASSES + 1].append(rmask >> 32)
        row_cols[ROW_ACTION_CLASSES].append(amask & 0xFFFFFFFF)
        row_cols[ROW_ACTION_CLASSES + 1].append(amask >> 32)
        row_cols[ROW_FLAGS][i] |= (ROW_F_ROLE_BY_CLASS if rexact else 0) | (ROW_F_ACTION_BY_CLASS if aexact else 0)
    # the leaf half: a copy of the condition's 8-dword fused-leaf record (celc.py condition_program), so that the
    # visit which needs the condition already has it; the derived-role condition's leaf likewise, in its own section
    leaf2_cols = [[] for _ in range(8)]
    for i, (cond, drc) in enumerate(zip(row_cols[ROW_COND], row_cols[ROW_DRCOND])):
        for ref, flag, tflag, cols, base in ((cond, ROW_F_LEAF_EMBEDDED, ROW_F_TREE_EMBEDDED, row_cols, ROW_LEAF),
                                             (drc, ROW_F_DRLEAF_EMBEDDED, ROW_F_DRTREE_EMBEDDED, leaf2_cols, 0)):
            rec = [0] * 8
            if ref != NONE and (ref & COND_LEAF):
                pc = ref & COND_PC_MASK
                rec = [int(w) & 0xFFFFFFFF for w in pb.code[pc:pc + 8]]
                row_cols[ROW_FLAGS][i] |= flag
            elif ref != NONE and (ref & COND_LEAFTREE) and (ref & COND_PC_MASK) in pb.tree_strips:
                rec = _tree_descriptor(pb.tree_strips[ref & COND_PC_MASK])
                row_cols[ROW_FLAGS][i] |= tflag
            for k in range(8):
                cols[base + k].append(rec[k])
    # derived-role definitions once more for the flat kernel: parent roles as a class mask, the condition's leaf inline
    drx_cols = [[] for _ in range(16)]
    drx_exact = True
    for i, parents in enumerate(dr_parents):
        mask = 0
        for role in parents:
            if role == "*":
                mask = 0xFFFFFFFFFFFFFFFF
            elif "*" in role or role not in lt.string_ids or int(role_class[lt.string_ids[role]]) >= 62:
                mask, drx_exact = 0xFFFFFFFFFFFFFFFF, False     # a glob / a role outside the classes: not decidable by class
            else:
                mask |= 1 << int(role_class[lt.string_ids[role]])
        if small_classes:
            mask |= (mask >> 63) << 31
        cond = dr_cols[3][i]
        rec = [0] * 8
        emb = cond != NONE and bool(cond & COND_LEAF)
        if emb:
            pc = cond & COND_PC_MASK
            rec = [int(w) & 0xFFFFFFFF for w in pb.code[pc:pc + 8]]
        tree = cond != NONE and bool(cond & COND_LEAFTREE) and (cond & COND_PC_MASK) in pb.tree_strips
        if tree:
            rec = _tree_descriptor(pb.tree_strips[cond & COND_PC_MASK])
        vals = [mask & 0xFFFFFFFF, mask >> 32, (1 if emb else 0) | (2 if tree else 0), cond, dr_cols[0][i], 0, 0, 0] + rec
        for k in range(16):
            drx_cols[k].append(vals[k])

    # ---- cbh_check_walk2.h: what the general decision kernel reads besides the records themselves
    ALL64 = 0xFFFFFFFFFFFFFFFF
    walk2_why = []     # reasons the table stays on the older kernels (stats["walk2_refused"])

    def class_bit(table, key):
        i = lt.string_ids.get(key)
        c = int(table[i]) if i is not None else 63
        return (1 << c) if c < 62 else None

    def x_masks(keys, table, dim, meta_is_glob=False):
        """(literal class mask, glob mask, exact?) of a role / action list for the kernel that matches by class AND by glob
        bit: '*' = every class; a glob = its bit (the lane holds the match bits of its strings for the first
        WALK2_MAX_GLOBS patterns of the dimension); a literal = its class."""
        lit, glob, exact = 0, 0, keys is not None
        for key in keys or ():
            if key == "*" and not meta_is_glob:
                lit = ALL64
            elif ("*" in key) or (meta_is_glob and has_meta(key)):
                gi = dims[dim].globs.get(key)
                if gi is None or gi >= WALK2_MAX_GLOBS:
                    exact = False
                else:
                    glob |= 1 << gi
            else:
                b = class_bit(table, key) if key else None
                if b is None:
                    exact = False
                else:
                    lit |= b
        return lit, glob, exact

    n_rows_total = len(row_cols[0])
    rowx = np.zeros((n_rows_total, 8), dtype=np.uint32)
    rowx[:, 0] = GSLOT_NONE | (GSLOT_NONE << 16)
    row_rmask = []   # per row (literal role class mask, role glob mask) for the buckets' union masks
    for i in range(n_rows_total):
        rl, rg, rex = x_masks(row_roles[i], role_class, DIM_ROLE)
        al, ag, aex = x_masks(row_actions[i], action_class, DIM_ACTION)
        row_rmask.append((rl, rg))
        rowx[i, 1] = ag | (rg << 16)
        rowx[i, 2], rowx[i, 3], rowx[i, 4], rowx[i, 5] = rl & 0xFFFFFFFF, rl >> 32, al & 0xFFFFFFFF, al >> 32
        if row_roles[i] is not None:
            if rex and aex:
                row_cols[ROW_FLAGS][i] |= ROW_F_XEXACT
            else:
                walk2_why.append("a rule's role / action list is not decidable by class and glob bits")
        if ag or rg:
            row_cols[ROW_FLAGS][i] |= ROW_F_X
    # role-policy rules once more in the form that kernel reads (CbhRpxField): allow list as class + glob masks, the
    # condition's leaf / tree descriptor inline
    n_rp_total = len(rp_cols[0])
    rpx = np.zeros((n_rp_total, 16), dtype=np.uint32)
    for i in range(n_rp_total):
        al, ag, aex = x_masks(rp_allow[i], action_class, DIM_ACTION, meta_is_glob=True)
        if not aex:
            walk2_why.append("a role policy's allow list is not decidable by class and glob bits")
        cond = rp_cols[3][i]
        rec, fl = [0] * 8, 0
        if cond != NONE and (cond & COND_LEAF):
            pc = cond & COND_PC_MASK
            rec, fl = [int(w) & 0xFFFFFFFF for w in pb.code[pc:pc + 8]], 1
        elif cond != NONE and (cond & COND_LEAFTREE) and (cond & COND_PC_MASK) in pb.tree_strips:
            rec, fl = _tree_descriptor(pb.tree_strips[cond & COND_PC_MASK]), 2
        if trace_rp_rules[i].get("emit_output"):
            fl |= 4     # CBH_RPX_HOW bit 2: the rule has output expressions
        rpx[i, :8] = [rp_cols[0][i], rp_cols[2][i], cond, GSLOT_NONE, al & 0xFFFFFFFF, al >> 32, ag, fl]
        rpx[i, 8:] = rec
    drx_gslot = [GSLOT_NONE] * len(drx_cols[0])

    # Evaluation sites and their slots.  What the walk cannot decide inline - a generic program, or a classified leaf that
    # meets an int / uint / container value - is evaluated by a launch of its own before the walk (cbh_walk2_pre_kernel: the
    # interpreter's registers stay out of the walk) which leaves two result bits per site and request.  A site's slot is
    # unique among everything ONE request can reach: the rules and derived roles of its (version, kind), the rules of its
    # principal's policies, the role-policy rules of its version.  Generic sites take the low slots: a batch of plain
    # scalars needs only those.
    def site_kind(ref, leaf_class, how):
        if ref == NONE:
            return None
        if how == 2 or (how == 1 and leaf_class in (1, 2, 3, 4, 6, 7, 8)):
            return "open"
        return "generic"

    # (A program that reads runtime.effectiveDerivedRoles sees the derived roles of the scope being walked, check.go:237-282:
    # with such programs in the table a slot belongs to one scope's site, not to the program.)
    per_scope = bool(pb.uses_runtime)
    sites = []   # (range, family, program [, scope], kind, setter)
    for i, f in enumerate(row_cols[ROW_FLAGS]):
        fam = row_family[i]
        how = 1 if f & ROW_F_LEAF_EMBEDDED else 2 if f & ROW_F_TREE_EMBEDDED else 0
        k = site_kind(row_cols[ROW_COND][i], row_cols[ROW_LEAF + 7][i], how)
        if k:
            sites.append((fam[0], fam, (row_cols[ROW_COND][i], row_scope[i] if per_scope else None), k, ("row", i, 0)))
        how = 1 if f & ROW_F_DRLEAF_EMBEDDED else 2 if f & ROW_F_DRTREE_EMBEDDED else 0
        k = site_kind(row_cols[ROW_DRCOND][i], leaf2_cols[7][i], how)
        if k:
            sites.append((fam[0], fam, (row_cols[ROW_DRCOND][i], row_scope[i] if per_scope else None), k, ("row", i, 16)))
    for i in range(len(drx_cols[0])):
        k = site_kind(drx_cols[3][i], drx_cols[15][i], drx_cols[2][i] & 3)
        if k:
            sites.append(("R", dr_family[i], (drx_cols[3][i], "dr"), k, ("dr", i, 0)))
    for i in range(n_rp_total):
        k = site_kind(int(rpx[i, 2]), int(rpx[i, 15]), int(rpx[i, 7]) & 3)
        if k:
            sites.append(("Q", ("Q", rp_family[i]), (int(rpx[i, 2]), rp_scope[i] if per_scope else None), k, ("rp", i, 0)))
    # the probes of the params sets' variables (celc.py vars_probe_program): generic sites of their own
    rowx[:, 6] = GSLOT_NONE | (GSLOT_NONE << 16)
    rowx[:, 7] = NONE
    for i, (pa, pd) in enumerate(row_probe):
        fam = row_family[i]
        if pa is not None:
            sites.append((fam[0], fam, (pa, row_scope[i] if per_scope else None), "generic", ("rowp", i, 0)))
        if pd is not None:
            sites.append((fam[0], fam, (pd, row_scope[i] if per_scope else None), "generic", ("rowp", i, 16)))
        if pa is not None or pd is not None:
            rowx[i, 7] = len(pool)
            pool.extend([pa if pa is not None else NONE, pd if pd is not None else NONE])
            row_cols[ROW_FLAGS][i] |= ROW_F_X
    for i, pr in enumerate(dr_probe):
        drx_cols[6][i] = pr if pr is not None else NONE
        if pr is not None:
            sites.append(("R", dr_family[i], (pr, "dr"), "generic", ("drp", i, 0)))
    for ei, (pr, ver, scope) in rp_bucket_probe.items():
        sites.append(("Q", ("Q", ver), (pr, scope if per_scope else None), "generic", ("rpp", ei, 0)))
    drx_pslot = [GSLOT_NONE] * len(drx_cols[0])
    slot_of, next_free, spans = {}, {}, {}
    for kind in ("generic", "open"):
        base = sum(spans.values())
        for rng in ("R", "P", "Q"):
            span = 0
            for (r, fam, prog, k, setter) in sites:
                if r != rng or k != kind:
                    continue
                key = (fam, prog)
                if key not in slot_of:
                    n = next_free.get((kind, fam), 0)
                    next_free[(kind, fam)] = n + 1
                    slot_of[key] = base + n
                    span = max(span, n + 1)
            spans[(kind, rng)] = span
            base += span
        if kind == "generic":
            n_gslots_generic = sum(spans.values())
    n_gslots_all = sum(spans.values())
    if n_gslots_all > WALK2_MAX_GSLOTS:
        walk2_why.append("more than %d evaluation sites on one request's path" % WALK2_MAX_GSLOTS)
    else:
        for (r, fam, prog, k, (what, i, shift)) in sites:
            g = slot_of[(fam, prog)]
            if what == "row":
                rowx[i, 0] = (int(rowx[i, 0]) & ~(0xFFFF << shift)) | (g << shift)
                row_cols[ROW_FLAGS][i] |= ROW_F_X
            elif what == "rowp":
                rowx[i, 6] = (int(rowx[i, 6]) & ~(0xFFFF << shift)) | (g << shift)
            elif what == "dr":
                drx_gslot[i] = g
            elif what == "drp":
                drx_pslot[i] = g
            elif what == "rpp":
                e = entries[i]
                entries[i] = e[:7] + (len(pool),)
                pool.extend([rp_bucket_probe[i][0], g])
            else:
                rpx[i, 3] = g
    for i, g in enumerate(drx_gslot):
        drx_cols[5][i] = g | (drx_pslot[i] << 16)

    # the roles with role policies at (version, scope), sorted by name: index.go:352-530 walks a request role's
    # [role] ++ ancestors list, and the ancestors come sorted (ruletable/build.py; the reference's own order is Go's map
    # iteration) - the kernel visits a slot's OWN role first, then this list
    rp_by_vs = {}
    for (ver, scope, role) in sorted(rp_buckets):
        rp_by_vs.setdefault((ver, scope), []).append(role)
    for (ver, scope), roles_here in sorted(rp_by_vs.items()):
        order = sorted(roles_here)
        if len(order) > 32:
            walk2_why.append("more than 32 role policies in one scope")
        for r in order:
            if class_bit(role_class, r) is None:
                walk2_why.append("a role with a role policy has no role class")
        entries.append((B_RPROLES, lt.string_ids[ver], lt.scope_index[scope], 0, len(pool), len(order), 0, 0))
        pool.extend(lt.string_ids[r] for r in order)
    # union of the role lists of a resource policy's rules (the base bitmap of Index.Query, index.go:250-305: a scope
    # yields role-policy DENYs only if some binding there ties the resource to one of the roles)
    bucket_union = {}
    for e in entries:
        if e[0] == B_RESOURCE:
            lit = glob = 0
            for i in range(e[4], e[4] + e[5]):
                lit |= row_rmask[i][0]; glob |= row_rmask[i][1]
            bucket_union[e[1:4]] = (lit, glob)
    # ... and which kinds of evaluation sites the bucket holds (the pre-pass skips what has none): v0 = 1 | 2 rules with
    # generic sites | 4 rules with sites that are inline for plain values | 8, 16 the same for its derived-role definitions
    bucket_sites = {}
    site_bit = {("row", "generic"): 2, ("row", "open"): 4, ("dr", "generic"): 8, ("dr", "open"): 16}
    row_key, dr_key = {}, {}
    for e in entries:
        if e[0] == B_RESOURCE:
            for i in range(e[4], e[4] + e[5]):
                row_key[i] = e[1:4]
            for i in range(e[6], e[6] + e[7]):
                dr_key[i] = e[1:4]
    for (_r, _fam, _prog, k, (what, i, _shift)) in sites:
        what = {"rowp": "row", "drp": "dr"}.get(what, what)
        key = row_key.get(i) if what == "row" else dr_key.get(i) if what == "dr" else None
        if key is not None:
            bucket_sites[key] = bucket_sites.get(key, 0) | site_bit[(what, k)]
    # a role's ancestors once more as what the walk needs of them: the OR of their classes (v2, v3) - one directory probe
    # instead of a walk over the list (which only a table with role globs still makes, for their match bits)
    def _parents_entry(e):
        lit = 0
        for anc_sid in pool[e[4]:e[4] + e[5]]:
            c = int(role_class[anc_sid])
            lit |= 1 << (c if c < 62 else 63)
        return (e[0], e[1], e[2], e[3], e[4], e[5], lit & 0xFFFFFFFF, lit >> 32)
    entries[:] = [_parents_entry(e) if e[0] == B_PARENTS else e for e in entries]
    family_sites = {}
    for (_v, _k, _s), bits in bucket_sites.items():
        family_sites[(_v, _k)] = family_sites.get((_v, _k), 0) | bits
    # ... and WHOSE requests can reach a generic site of the family: the union of the role classes of the records / definitions
    # that carry one (v1, v2; v3 bit 0: usable - no role glob among them).  The pre-pass lets a request whose role sets miss it
    # go (cbh_check_walk2.h): requests are grouped by route and role list, so whole waves leave before the walk.
    family_roles = {}
    ALL = 0xFFFFFFFFFFFFFFFF
    for (_r, _fam, _prog, k, (what, i, _shift)) in sites:
        if k != "generic":
            continue
        what = {"rowp": "row", "drp": "dr"}.get(what, what)
        if what == "row" and i in row_key:
            lit, glob = row_rmask[i]
            m = ALL if glob or row_roles[i] is None else lit
            key = row_key[i]
        elif what == "dr" and i in dr_key:
            m = drx_cols[0][i] | (drx_cols[1][i] << 32)
            key = dr_key[i]
        else:
            continue
        family_roles[key[:2]] = family_roles.get(key[:2], 0) | m
    for (fv, fk) in sorted({(e[1], e[2]) for e in entries if e[0] == B_RESOURCE}):
        fm = family_roles.get((fv, fk), 0)
        entries.append((B_FAMILY, fv, fk, 0, family_sites.get((fv, fk), 0), fm & 0xFFFFFFFF, fm >> 32, 0 if fm == ALL else 1))
    # per table string: is it a principal with a principal policy, a role with ancestors in some scope (what the walk would
    # otherwise probe the directory for, lane by lane)
    str_wflags = np.zeros(K, dtype=np.uint8)
    for (_v, _s, principal) in prin_buckets:
        str_wflags[lt.string_ids[principal]] |= SWF_PRINCIPAL
    for _scope, roles in rt["parent_roles"].items():
        for role, ancestors in roles.items():
            if ancestors:
                str_wflags[lt.string_ids[role]] |= SWF_PARENTS
    for (_v, scope, _r) in rp_buckets:
        scope_flags[lt.scope_index[scope]] |= SCOPE_F_ROLEPOL
    q_sites = 0   # the role-policy rules' sites (any request of the version can reach them): 2 generic, 4 open
    for (r, _fam, _prog, k, _setter) in sites:
        if r == "Q":
            q_sites |= 2 if k == "generic" else 4
    entries[:] = [((e[0], e[1], e[2], e[3], 1 | bucket_sites.get(e[1:4], 0)) + (lambda u: (u[0] & 0xFFFFFFFF, u[0] >> 32, u[1]))(bucket_union.get(e[1:4], (0, 0)))) if e[0] == B_RESEXISTS else e
                  for e in entries]

    # ---- assemble
    lt.columns = [k for k, _ in sorted(pb.columns.items(), key=lambda kv: kv[1])]
    lt.per_call_globals = pb.per_call_globals
    lt.dr_names = [n for n, _ in sorted(pb.dr_names.items(), key=lambda kv: kv[1])]
    lt.unsupported = list(dict.fromkeys(pb.unsupported))
    assert len(lt.strings) == K, "string interned after the pool was frozen"

    enc = [s.encode("utf-8") for s in lt.strings]
    str_off = np.zeros(K + 1, dtype=np.uint32)
    str_off[1:] = np.cumsum([len(b) for b in enc])
    meta = np.zeros(META_N, dtype=np.uint32)
    meta[M_NSTRINGS] = K
    meta[M_NCOLUMNS] = len(lt.columns)
    meta[M_NSCOPES] = len(lt.scopes)
    meta[M_NROWS] = len(row_cols[0])
    meta[M_NRPROWS] = len(rp_cols[0])
    meta[M_NDR] = len(dr_cols[0])
    meta[M_NPOLICIES] = len(lt.policy_keys)
    meta[M_NCONSTS] = len(pb.const_tag)
    meta[M_CODE_LEN] = len(pb.code)
    meta[M_FLAGS] = ((MF_USES_RUNTIME_EDR if pb.uses_runtime else 0)
                     | (2 if any(v for r in rt["parent_roles"].values() for v in r.values()) else 0)
                     | (4 if rp_buckets else 0)
                     | (8 if pb.has_generic else 0)
                     | (16 if used_any else 0)
                     | (32 if pp_exists else 0)
                     | (64 if any(f >= 10 for f in pb.req_fields) else 0)
                     | (128 if (pb.reads_string_bytes or any(lt.nfas[d].patterns for d in range(3))) else 0))
    # FLAT (cbh_check_flat.h): nothing but resource policies with leaf conditions whose records the class masks decide
    max_depth = max(len(list(namer.scope_parents(sc))) + 1 if sc else 1 for sc in lt.scopes)
    by_class = ROW_F_ROLE_BY_CLASS | ROW_F_ACTION_BY_CLASS
    flat = (small_classes and not pb.has_generic and not pb.uses_runtime and drx_exact and len(lt.dr_names) <= 64 and not rp_buckets and not pp_exists
            and not (int(meta[M_FLAGS]) & 2) and not any(lt.nfas[d].patterns for d in range(3)) and max_depth <= 16
            and all((f & by_class) == by_class for f in row_cols[ROW_FLAGS]))
    meta[M_FLAGS] |= 256 if flat else 0
    # FLAT_CLOSED: besides, every condition is a classified leaf or a tree of them the flat kernel evaluates inline -
    # with a batch of plain scalars no evaluation can need the shared evaluator (the kernel variant without the call)
    def inline_ok(ref, leaf_slot_class, embedded, tree):
        if tree:   # every leaf of the strip must be one the kernels without the evaluator call decide (the mask walk's blocks: classes 1-4, 6)
            _packed, n, first = pb.tree_strips[ref & COND_PC_MASK]
            return all(pb.code[(first + j) * 8 + 7] in (1, 2, 3, 4, 6) for j in range(n))
        return ref == NONE or (embedded and leaf_slot_class in (1, 2, 3, 4, 6))
    closed = flat and all(
        inline_ok(row_cols[ROW_COND][i], row_cols[ROW_LEAF + 7][i], f & ROW_F_LEAF_EMBEDDED, f & ROW_F_TREE_EMBEDDED)
        and inline_ok(row_cols[ROW_DRCOND][i], leaf2_cols[7][i], f & ROW_F_DRLEAF_EMBEDDED, f & ROW_F_DRTREE_EMBEDDED)
        for i, f in enumerate(row_cols[ROW_FLAGS])) and all(
        inline_ok(drx_cols[3][i], drx_cols[15][i], drx_cols[2][i] & 1, drx_cols[2][i] & 2) for i in range(len(drx_cols[0])))
    meta[M_FLAGS] |= MF_FLAT_CLOSED if closed else 0
    # ---- segments (cbh_check_flat.h, the mask walk): a flat table's buckets once more, 64 records at a time, as the
    # reference's own index holds them - one bitmap per dimension value (index/index.go:270-305, bitmap.go:109-158)
    seg_words, leaf_pool, n_pool_leaves = [], [], 0
    if flat:
        seg_words, leaf_pool, seg_entries, seg_stats = _segments(
            [e for e in entries if e[0] == B_RESOURCE], row_cols, leaf2_cols, pb)
        entries.extend(seg_entries)
        n_pool_leaves = seg_stats["pool_slots"]
        meta[M_SEGS] = M_SEGS_PRESENT | (M_SEGS_POOLED if seg_stats["pooled"] else 0) | n_pool_leaves
        lt.seg_stats = seg_stats

    # ---- directory hash table
    nslots = 16
    while nslots < 2 * len(entries):
        nslots *= 2
    slots = np.full((nslots, 8), NONE, dtype=np.uint32)
    for e in entries:
        i = hash4(e[0], e[1], e[2], e[3]) & (nslots - 1)
        while slots[i, 0] != NONE:
            if tuple(slots[i, :4]) == e[:4]:
                raise LoweringError("duplicate directory key %r" % (e[:4],))
            i = (i + 1) & (nslots - 1)
        slots[i, :] = e
    meta[M_HASH_MASK] = nslots - 1
    meta[M_FLAGS] |= 1024 if pb.needs_arena else 0   # CBH_MF_NEEDS_ARENA: the interpreter kernels get the lanes' list arenas
    # WALK2 (cbh_check_walk2.h): every record decided by class masks + glob bits, derived roles by class, no program that
    # reads runtime.effectiveDerivedRoles (its value is the scope's being walked: the older kernel keeps those tables)
    if not drx_exact or len(lt.dr_names) > 64:
        walk2_why.append("a derived role's parent roles are not decidable by class")
    if max_depth > 16:
        walk2_why.append("scope chains longer than 16")
    for d in (DIM_ACTION, DIM_ROLE):
        if len(dims[d].globs) > WALK2_MAX_GLOBS:
            walk2_why.append("more than %d glob patterns in a dimension" % WALK2_MAX_GLOBS)
    if lt.trace_has_variables or lt.trace_has_outputs:
        meta[M_FLAGS] |= MF_TRACE_ALL
    lt.walk2_refused = list(dict.fromkeys(walk2_why))
    meta[M_FLAGS] |= MF_WALK2 if not lt.walk2_refused else 0
    lt.inline_cols = sorted(pb.inline_cols)
    meta[M_INLINE_COLS] = (max(pb.inline_cols) + 1) if pb.inline_cols else 0
    meta[M_SENS_COLS] = sum(1 << c for c in pb.sensitive_cols if c < 32)
    meta[M_GSLOTS_GENERIC] = n_gslots_generic
    meta[M_GSLOTS_ALL] = n_gslots_all
    meta[M_Q_SITES] = q_sites | (8 if any(r == "P" and k == "generic" for (r, _f, _p, k, _s) in sites) else 0) \
        | (16 if any(r == "P" and k == "open" for (r, _f, _p, k, _s) in sites) else 0)
    meta[M_MAX_STACK] = pb.max_stack
    meta[M_NDRNAMES] = len(lt.dr_names)
    meta[M_NFA_WORDS_ACTION] = lt.nfas[0].words
    meta[M_NFA_WORDS_ROLE] = lt.nfas[1].words
    meta[M_NFA_WORDS_KIND] = lt.nfas[2].words
    meta[M_THEAP_LEN] = len(pb.theap_tag)
    meta[M_MAX_LOCALS] = pb.max_locals

    def u32(a):
        return np.asarray(a, dtype=np.uint32).tobytes()

    def u64(a):
        return np.array([int(x) & 0xFFFFFFFFFFFFFFFF for x in a], dtype=np.uint64).tobytes()

    def u8(a):
        return np.asarray(a, dtype=np.uint8).tobytes()

    def val_records(tags, vals):
        out = np.zeros((len(tags), 4), dtype=np.uint32)
        for i, (tg, v) in enumerate(zip(tags, vals)):
            v = int(v) & 0xFFFFFFFFFFFFFFFF
            out[i] = (tg, 0, v & 0xFFFFFFFF, v >> 32)
        return out.tobytes()

    def row_major(cols, width):
        """Records of `width` u32 (fields = cols, zero padded): one wide scalar load per record."""
        n = len(cols[0])
        out = np.zeros((n, width), dtype=np.uint32)
        for k, c in enumerate(cols):
            out[:, k] = np.asarray(c, dtype=np.uint32) if n else []
        return out.tobytes()

    code = list(pb.code) or [celc.OP_RET]
    sections = [
        (SEC_META, META_N, meta.tobytes()),
        (SEC_STR_OFF, K + 1, str_off.tobytes()),
        (SEC_STR_BYTES, int(str_off[-1]), b"".join(enc)),
        (SEC_SCOPE_PARENT, len(lt.scopes), u32(scope_parent)),
        (SEC_SCOPE_FLAGS, len(lt.scopes), u32(scope_flags)),
        (SEC_SCOPE_SID, len(lt.scopes), u32(scope_sid)),
        (SEC_HASH, nslots, slots.tobytes()),
        (SEC_ROWS, len(row_cols[0]), row_major(row_cols, 16)),
        (SEC_ROWPAT, len(pat_cols[0]), row_major(pat_cols, 8)),
        (SEC_ROWLEAF2, len(leaf2_cols[0]), row_major(leaf2_cols, 8)),
        (SEC_DRX, len(drx_cols[0]), row_major(drx_cols, 16)),
        (SEC_ROWX, n_rows_total, rowx.tobytes()),
        (SEC_RPX, n_rp_total, rpx.tobytes()),
        (SEC_STR_WFLAGS, K, str_wflags.tobytes()),
        (SEC_SEGS, len(seg_words) // 16, u32(seg_words + [0] * (16 * SEG_TAIL_PAD))),   # (+ a tail the speculative staging loads may read)
        (SEC_LEAFPOOL, n_pool_leaves, u32(leaf_pool + [0] * 16)),
        (SEC_REGEX, len(pb.regex_words), u32(pb.regex_words)),   # DFA tables of constant `matches` patterns (lower/regex.py)
        (SEC_RPROWS, len(rp_cols[0]), row_major(rp_cols, 4)),
        (SEC_U32POOL, len(pool), u32(pool)),
        (SEC_DR, len(dr_cols[0]), row_major(dr_cols, 4)),
        (SEC_CODE, len(code), u32(code)),
        (SEC_CONST_TAG, len(pb.const_tag), u8(pb.const_tag)),
        (SEC_CONST_VAL, len(pb.const_val), u64(pb.const_val)),
        (SEC_THEAP_TAG, len(pb.theap_tag), u8(pb.theap_tag)),
        (SEC_THEAP_VAL, len(pb.theap_val), u64(pb.theap_val)),
        (SEC_GBITS, 3 * K, gbits.tobytes()),
        (SEC_NFA_ACTION, lt.nfas[0].words, nfa_bytes[0]),
        (SEC_NFA_ROLE, lt.nfas[1].words, nfa_bytes[1]),
        (SEC_NFA_KIND, lt.nfas[2].words, nfa_bytes[2]),
        (SEC_POLICY_SID, len(lt.policy_keys), u32([0] * len(lt.policy_keys))),
        (SEC_DRNAME_SID, len(lt.dr_names), u32([0] * len(lt.dr_names))),
        # the same constants / constant-heap entries as 16-byte records {tag, 0, lo, hi}: one
        # s_load_dwordx4 at a wave-uniform index (fused leaves read their constant operands this way)
        (SEC_CONST_REC, len(pb.const_tag), val_records(pb.const_tag, pb.const_val)),
        (SEC_THEAP_REC, len(pb.theap_tag), val_records(pb.theap_tag, pb.theap_val)),
        (SEC_ROLE_CLASS, K, role_class.tobytes()),
        (SEC_ACTION_CLASS, K, action_class.tobytes()),
        # host only: where each attribute column comes from (the C++ ingest walks these paths)
        (SEC_COLUMN_PATHS, len(lt.columns), _column_paths(lt.columns)),
        # host only: policy keys of CBH_P_TABLE policy words, then derived-role names in edr_mask bit order
        (SEC_HOST_NAMES, len(lt.policy_keys) + len(lt.dr_names), _names(lt.policy_keys) + _names(lt.dr_names)),
    ]
    if trace:
        def records(rows, width):
            return np.asarray(rows, dtype=np.uint32).reshape(len(rows), width).tobytes()
        # host only, JSON: what a consumer of the trace records needs besides the batch it sent - the strings the records
        # refer to (expression texts, variable names, rule FQNs) and the templates of the output expressions
        # (celc.py _output_template, keyed by the records' rule word)
        import json
        tstr = json.dumps({"strings": lt.trace_strings,
                           "templates": {str(k): [t, n] for k, (t, n) in sorted(lt.trace_templates.items())}},
                          separators=(",", ":")).encode("utf-8")
        sections += [
            (SEC_TRACE_ROWS, len(trace_rows), records(trace_rows, 8)),
            (SEC_TRACE_DR, len(trace_dr), records(trace_dr, 4)),
            (SEC_TRACE_RP, len(trace_rp), records(trace_rp, 8)),
            (SEC_TRACE_POOL, len(trace_pool), u32(trace_pool or [0])),
            (SEC_TRACE_STRINGS, len(lt.trace_strings), tstr),   # host only
            (SEC_TRACE_HOST, len(lt.trace_strings), _trace_host(lt.trace_strings, lt.trace_templates)),   # host only: the same for cbh_ingest.cpp
        ]
    lt.blob = _pack(sections)
    lt.stats = {
        "strings": K, "scopes": len(lt.scopes), "rows": len(row_cols[0]), "role_policy_rows": len(rp_cols[0]),
        "derived_roles": len(dr_cols[0]), "programs": len(pb.programs), "code_words": len(pb.code),
        "columns": len(lt.columns), "directory_slots": nslots, "blob_bytes": len(lt.blob),
        "unsupported_expressions": len(lt.unsupported),
        "globs": [len(d.globs) for d in dims],
        "reads_request_strings": bool(int(meta[M_FLAGS]) & 64),
        "needs_string_bytes": bool(int(meta[M_FLAGS]) & 128),
        "walk2": bool(int(meta[M_FLAGS]) & MF_WALK2), "walk2_refused": lt.walk2_refused,
        "gslots": (int(n_gslots_generic), int(n_gslots_all)),
        "flat_closed": bool(int(meta[M_FLAGS]) & MF_FLAT_CLOSED),   # ... and every condition inline: the variant without the evaluator call serves plain batches
        "flat": bool(int(meta[M_FLAGS]) & 256),   # eligible for cbh_check_flat_kernel (batch shape and mode permitting)   # glob automata or programs that look inside strings   # raw request strings (R.id, R.kind, scopes, versions) read by some program
        "generic_programs": bool(pb.has_generic),   # selects the kernel with the operand-stack interpreter
        # feature class of the 32-bit-mask kernels (cbh_pick_check_kernel): "" = everything (role policies /
        # parent roles), else "_f<bits>" with bit 0 = derived roles, bit 2 = glob patterns
        "kernel_features": ("" if int(meta[M_FLAGS]) & (2 | 4 | 32) else "_f%d" % (
            (1 if (len(dr_cols[0]) or int(meta[M_FLAGS]) & MF_USES_RUNTIME_EDR) else 0)
            | (4 if (used_any or any(lt.nfas[d].patterns for d in range(3))) else 0))),
    }
    return lt


def _trace_host(strings, templates):
    """CBH_SEC_TRACE_HOST (cbh_blob.h): the trace strings and output templates in a form C++ reads without a JSON parser."""
    out = bytearray(struct.pack("<I", len(strings)))
    for x in strings:
        b = x.encode("utf-8")
        out += struct.pack("<I", len(b)) + b

    def node(t):
        k = t[0]
        if k == "hole":
            return struct.pack("<BI", 0, t[1])
        if k == "const":
            v = t[1]
            if v is None:
                return struct.pack("<BB", 1, 0)
            if isinstance(v, bool):
                return struct.pack("<BBB", 1, 1, int(v))
            if isinstance(v, int):
                return struct.pack("<BBq", 1, 2, v if v < (1 << 63) else v - (1 << 64))
            if isinstance(v, float):
                return struct.pack("<BBd", 1, 3, v)
            b = v.encode("utf-8")
            return struct.pack("<BBI", 1, 4, len(b)) + b
        if k == "list":
            return struct.pack("<BI", 2, len(t[1])) + b"".join(node(e) for e in t[1])
        if k == "map":
            return struct.pack("<BI", 3, len(t[1])) + b"".join(node(kt) + node(vt) for kt, vt in t[1])
        fb = t[1].encode("utf-8")
        return struct.pack("<BI", 4, len(fb)) + fb + struct.pack("<I", len(t[2])) + b"".join(node(e) for e in t[2])

    out += struct.pack("<I", len(templates))
    for word, (tmpl, n_holes) in sorted(templates.items()):
        out += struct.pack("<II", word, n_holes) + node(tmpl)
    return bytes(out)


def _tree_descriptor(strip):
    """The leaf slot of a record whose condition is a tree of classified leaves (celc.py _tree_strip), laid over the
    fused-leaf record's fields: {ops 0-7, ops 8-15, index of the first leaf record in 8-dword units of the tape,
    ops 16-23, ops 24-31, number of leaves, 0, 7}."""
    packed, n, first = strip
    return [packed[0], packed[1], first, packed[2], packed[3], n, 0, 7]


def _layout_leaves(leaves):
    """Fused-leaf records (8 dwords, celc.py _leaf_record) -> (slot of each leaf, the words of their 4-dword forms, slots).

    The mask walk evaluates a pool front to back, FOUR leaves (one 16-dword scalar load) at a time, and the four of a block
    are of one class - the block's code is straight-line, no per-leaf dispatch (cbh_check_flat.h eval_leaf_pool): the leaves
    are laid out by (class, column) and every class's run is padded to a multiple of four with copies of its last leaf.
    CbhLeaf4 (cbh_blob.h): {class | op << 4 | column << 16 | second column << 24, c0, c1, c2} - class 1: c0 = the constant's
    tag, c1 = its low dword; class 2: c0, c1 = the double's dwords; class 6: c0..c2 = the (at most three) string ids; class 3:
    both columns; class 4: the column compared with P.id; class 0 = none of these (the lane goes to the shared evaluator)."""
    def compact(lf):
        w, a0, a1, _ret, ctag, clo, chi, cls = lf
        a = w >> 8
        op, ka = a & 0xFF, (a >> 8) & 0xF
        if cls in (1, 6):
            rec = (cls | (op << 4) | (a0 << 16), ctag, clo, chi)
        elif cls == 2:
            rec = (cls | (op << 4) | (a0 << 16), clo, chi, 0)
        elif cls == 3:
            rec = (cls | (op << 4) | (a0 << 16) | (a1 << 24), 0, 0, 0)
        elif cls == 4:
            rec = (cls | (op << 4) | ((a0 if ka == 3 else a1) << 16), 0, 0, 0)
        else:
            return (0, 0, 0, 0)
        if max(a0 if cls != 4 or ka == 3 else a1, a1 if cls == 3 else 0) > 0xFF:
            raise LoweringError("attribute column beyond 255 in a fused leaf")
        return rec
    recs = sorted(((compact(lf), lf) for lf in leaves), key=lambda t: (t[0][0] & 15, (t[0][0] >> 16) & 0xFFFF, t[0]))
    slot_of, words = {}, []
    k = 0
    while k < len(recs):
        cls = recs[k][0][0] & 15
        run = [r for r in recs[k:] if (r[0][0] & 15) == cls]
        k += len(run)
        for rec, lf in run:
            slot_of[lf] = len(words) // 4
            words += list(rec)
        for _ in range(-len(run) % 4):
            words += list(run[-1][0])
    return slot_of, words, len(words) // 4


def _segments(res_entries, row_cols, leaf2_cols, pb):
    """The buckets of a FLAT table as SEGMENTS for the flat kernel's mask walk (cbh_check_flat.h, cbh_blob.h CBH_SEC_SEGS).

    A segment holds up to 64 consecutive records of one bucket as bit masks - bit i = record i of the segment, in binding
    order: per action class and per role class the records whose list holds it (what `Index.Query` ANDs per request,
    index/index.go:270-305), the ALLOW and the DENY records, and per DISTINCT condition of the segment (an item) the records
    it is the condition / the derived-role condition of.  A request's candidates are then (OR of its actions' masks) AND (OR
    of its roles' masks), a condition is evaluated once per wave and request whatever the number of records that carry it,
    and the first DENY of a walk is a count-trailing-zeros - no record is visited.

    A table whose conditions are built from at most 64 distinct fused leaves is POOLED: the leaves are numbered table-wide and
    live once in CBH_SEC_LEAFPOOL, and what a wave evaluated for one bucket serves the next; with at most 64 distinct
    conditions the items are numbered table-wide too (ITEMS_POOLED).
    Returns (words of the section, leaf pool, directory entries, stats)."""
    NONE_ = 0xFFFFFFFF

    def one_level(packed):
        """A tree's 4-bit ops (celc.py _tree_strip) -> (any?, negated?) when it is, after dropping levels that hold a single
        subtree, ONE level over leaves only; None otherwise.  all(x) = any(x) = x and none(x) = not x with x's errors kept
        (a level's only child is always evaluated: check.go:697-749) - which is what ruletable.go wraps a condition in for
        a REQUIRE_PARENTAL_CONSENT scope (none(cond))."""
        ops = []
        for k in range(32):
            o = (packed[k // 8] >> (4 * (k % 8))) & 15
            if o == 0:
                break
            ops.append(o)
        pos = [0]

        def node():
            kind = ops[pos[0]] - 2
            pos[0] += 1
            children = []
            while ops[pos[0]] < 8:
                if ops[pos[0]] == 1:
                    children.append("leaf")
                    pos[0] += 1
                elif 2 <= ops[pos[0]] < 5:
                    children.append(node())
                if 5 <= ops[pos[0]] < 8:
                    pos[0] += 1      # the child's TREE_ACC
            pos[0] += 1              # TREE_END
            return (kind, children)
        try:
            top = node()
            if pos[0] != len(ops):
                return None
        except IndexError:
            return None
        neg = False
        while len(top[1]) == 1 and top[1][0] != "leaf":
            neg ^= top[0] == 2
            top = top[1][0]
        if not top[1] or any(c != "leaf" for c in top[1]):
            return None
        kind = top[0]
        if len(top[1]) == 1:       # one leaf: all(l) = any(l) = l
            return (False, neg ^ (kind == 2))
        return (kind != 0, neg ^ (kind == 2))

    def item_of(ref, flags, leaf_rec, leaf_flag, tree_flag):
        """(how, leaves, ops): how & 3 = 1 - ONE level of classified leaves, any order (bit 8: any instead of all, bit 9:
        negated; a single fused leaf is all(leaf)), 2 = a deeper tree of classified leaves (its 4-bit ops), 0 = neither (the
        kernel hands those lanes to the shared evaluator, or flags them)."""
        if flags & leaf_flag:
            return 1, [tuple(leaf_rec)], [0, 0, 0, 0]
        if flags & tree_flag:
            d = leaf_rec
            first, n = d[2], d[5]
            leaves = [tuple(int(w) & 0xFFFFFFFF for w in pb.code[(first + j) * 8:(first + j) * 8 + 8]) for j in range(n)]
            ops = [d[0], d[1], d[3], d[4]]
            lvl = one_level(ops)
            if lvl is not None:
                return 1 | (256 if lvl[0] else 0) | (512 if lvl[1] else 0), leaves, [0, 0, 0, 0]
            return 2, leaves, ops
        return 0, [], [0, 0, 0, 0]

    n_rows = len(row_cols[0])
    row_items = []   # per row: [(is_drcond, ref, how, leaves, ops)]
    item_ref = {}
    all_items, all_leaves = {}, {}
    for i in range(n_rows):
        f = row_cols[ROW_FLAGS][i]
        its = []
        for is_dr, ref, rec, lf, tf in ((False, row_cols[ROW_COND][i], [row_cols[ROW_LEAF + k][i] for k in range(8)], ROW_F_LEAF_EMBEDDED, ROW_F_TREE_EMBEDDED),
                                        (True, row_cols[ROW_DRCOND][i], [leaf2_cols[k][i] for k in range(8)], ROW_F_DRLEAF_EMBEDDED, ROW_F_DRTREE_EMBEDDED)):
            if ref == NONE_:
                continue
            how, leaves, ops = item_of(ref, f, rec, lf, tf)
            # one item per distinct FUNCTION: programs of different policies that read the same (their params sets differ, their
            # fused leaves do not) share an item; `ref` stays a program that computes it (the shared evaluator's entry)
            key = (how, tuple(leaves), tuple(ops)) if how else ref
            its.append((is_dr, item_ref.setdefault(key, ref), how, leaves, ops))
        row_items.append(its)
    in_buckets = set()
    for e in res_entries:
        in_buckets.update(range(e[4], e[4] + e[5]))
    for i in sorted(in_buckets):
        for (_d, ref, _h, leaves, _o) in row_items[i]:
            all_items.setdefault(ref, len(all_items))
            for lf in leaves:
                all_leaves.setdefault(lf, len(all_leaves))
    pool_slot, pool_words, pool_slots = _layout_leaves(list(all_leaves))
    pooled = pool_slots <= SEG_RECORDS                              # the leaves are numbered table-wide
    words, entries = [], []
    n_segments = max_items = max_leaves = n_complex_total = 0

    def is_simple(how, ls):
        return (how & 3) == 1 and 1 <= len(ls) <= 4

    for e in res_entries:
        _t, ver, kind, scope_ix, begin, count, dr_begin, dr_count = e
        first_block = len(words) // 16
        n_seg = 0
        i = begin
        while i < begin + count:
            items, leaves, n_cx = {}, {}, 0       # ref -> local index; leaf record -> local index; complex items so far
            j = i
            while j < begin + count and j - i < SEG_RECORDS:
                new_items = {ref: (how, ls) for (_d, ref, how, ls, _o) in row_items[j] if ref not in items}
                new_leaves = {lf for (_d, _r, _h, ls, _o) in row_items[j] for lf in ls} - set(leaves)
                new_cx = sum(1 for (how, ls) in new_items.values() if not is_simple(how, ls))
                if (len(items) + len(new_items) > SEG_RECORDS or n_cx + new_cx > SEG_COMPLEX
                        or (not pooled and new_leaves and _layout_leaves(list(leaves) + list(new_leaves))[2] > SEG_RECORDS)):
                    break
                n_cx += new_cx
                for (_d, ref, _h, ls, _o) in row_items[j]:
                    items.setdefault(ref, len(items))
                    for lf in ls:
                        leaves.setdefault(lf, len(leaves))
                j += 1
            if j == i:
                raise LoweringError("a rule's conditions hold more than %d distinct leaves" % SEG_RECORDS)
            seg_slot, seg_leaf_words, seg_slots = ({}, [], 0) if pooled else _layout_leaves(list(leaves))
            lid = pool_slot if pooled else seg_slot
            allow = deny = simple_c = simple_d = 0
            am, rm = [0] * 32, [0] * 32
            item_recs = {}   # ref -> [crec_c, crec_d, how, leaves, ops]
            rec_c, rec_d = [0xFF] * 64, [0xFF] * 64
            for k, row in enumerate(range(i, j)):
                bit = 1 << k
                eff = row_cols[ROW_FLAGS][row] & 3
                allow |= bit if eff == 1 else 0
                deny |= bit if eff == 2 else 0
                a_lo, r_lo = row_cols[ROW_ACTION_CLASSES][row], row_cols[ROW_ROLE_CLASSES][row]
                for c in range(32):
                    if (a_lo >> c) & 1:
                        am[c] |= bit
                    if (r_lo >> c) & 1:
                        rm[c] |= bit
                for (is_dr, ref, how, ls, ops) in row_items[row]:
                    it = item_recs.setdefault(ref, [0, 0, how, ls, ops])
                    it[1 if is_dr else 0] |= bit
                    if is_simple(how, ls):
                        if is_dr:
                            rec_d[k], simple_d = items[ref], simple_d | bit
                        else:
                            rec_c[k], simple_c = items[ref], simple_c | bit
            ordered = sorted(items.items(), key=lambda kv: kv[1])
            descs, refs, complex_words = [], [], []
            for ref, _local in ordered:
                cc, cd, how, ls, ops = item_recs[ref]
                ids = [lid[lf] for lf in ls]
                refs.append(ref)
                if is_simple(how, ls):
                    b4 = ids + [0] * (4 - len(ids))
                    descs += [sum(b4[q] << (8 * q) for q in range(4)), len(ids) | (8 if how & 256 else 0) | (16 if how & 512 else 0)]
                else:
                    descs += [0, 0]
                    ids8 = ids + [0] * (8 - len(ids))
                    complex_words += [cc & 0xFFFFFFFF, cc >> 32, cd & 0xFFFFFFFF, cd >> 32, how, ref, items[ref], len(ls),
                                      sum(ids8[q] << (8 * q) for q in range(4)), sum(ids8[4 + q] << (8 * q) for q in range(4)), 0, 0] + list(ops)
            n_leaves = seg_slots
            block = [allow & 0xFFFFFFFF, allow >> 32, deny & 0xFFFFFFFF, deny >> 32, simple_c & 0xFFFFFFFF, simple_c >> 32,
                     simple_d & 0xFFFFFFFF, simple_d >> 32, len(items), n_leaves, 0, j - i, i, len(complex_words) // 16, 0, 0]
            for m in am + rm:
                block += [m & 0xFFFFFFFF, m >> 32]
            for tab in (rec_c, rec_d):
                block += [sum(tab[4 * q + z] << (8 * z) for z in range(4)) for q in range(16)]
            assert len(block) == SEG_FIXED_DWORDS
            block += descs
            off_refs = len(block)
            block += refs
            block += [0] * (-len(block) % 16)
            off_complex = len(block)
            block += complex_words
            off_leaves = len(block)
            block += seg_leaf_words
            block += [0] * (-len(block) % 16)
            block[10] = len(block) // 16
            block[14] = off_refs | (off_complex << 16)
            block[15] = off_leaves
            if len(block) >= 1 << 16:
                raise LoweringError("a segment block exceeds 2^16 dwords")
            words += block
            n_seg += 1
            n_segments += 1
            n_complex_total += len(complex_words) // 16
            max_items, max_leaves = max(max_items, len(items)), max(max_leaves, len(leaves))
            i = j
        entries.append((B_RESSEG, ver, kind, scope_ix, first_block, n_seg, dr_begin, dr_count))
    pool = pool_words if pooled else []
    hows = {}
    for its in row_items:
        for (_d, ref, how, _l, _o) in its:
            hows[ref] = how & 3
    stats = {"segments": n_segments, "pooled": pooled, "pool_slots": pool_slots if pooled else 0, "items": len(all_items),
             "items_one_level": sum(1 for r in all_items if hows[r] == 1), "items_deeper": sum(1 for r in all_items if hows[r] == 2),
             "complex_entries": n_complex_total, "leaves": len(all_leaves),
             "max_items": max_items, "max_leaves": max_leaves, "bytes": len(words) * 4}
    return words, pool, entries, stats


def _column_paths(columns):
    """Per column: u8 root (0 = P.attr, 1 = R.attr, 2 = auxData.jwt, 3 = auxData.jwts, 4 = the call's globals), u8 n_keys, then per key
    u16 length + UTF-8 bytes."""
    out = bytearray()
    for root, keys in columns:
        out += struct.pack("<BB", "PRJSG".index(root), len(keys))
        for k in keys:
            kb = k.encode("utf-8")
            out += struct.pack("<H", len(kb)) + kb
    return bytes(out)


def _names(names):
    """u32 count, then per name u16 length + UTF-8 bytes."""
    out = bytearray(struct.pack("<I", len(names)))
    for n in names:
        nb = n.encode("utf-8")
        out += struct.pack("<H", len(nb)) + nb
    return bytes(out)


def _pack(sections):
    hdr_len = 32 + 32 * len(sections)
    off = (hdr_len + 63) // 64 * 64
    table, bodies = [], []
    for sid_, count, data in sections:
        # keep every section non-empty so device pointers are always valid
        if not data:
            data = b"\0" * 8
        table.append(struct.pack("<IIQQQ", sid_, count, off, len(data), 0))
        pad = (-len(data)) % 64
        bodies.append(data + b"\0" * pad)
        off += len(data) + pad
    total = off
    head = struct.pack("<IIIIQQ", BLOB_MAGIC, BLOB_VERSION, len(sections), 0, total, 0)
    out = head + b"".join(table)
    out += b"\0" * ((-len(out)) % 64)
    return out + b"".join(bodies)
