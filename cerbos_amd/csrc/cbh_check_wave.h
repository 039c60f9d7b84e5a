// cbh_check_kernel - the decision kernel: request-major, wave-cooperative.
//
// One lane per CheckInput (principal, resource, up to 64 actions) restating
// ruletable.(*RuleTable).check (internal/ruletable/check.go:97-460).  Per-action state of the
// effect fold lives in 64-bit masks (bit k = k-th action of the request), so everything the
// reference does per action - rule match, ALLOW/DENY bookkeeping, scope permissions, the role
// fold - is bit-parallel over the request's actions, and a rule's condition is evaluated once per
// request instead of once per (action, rule).
//
// The TABLE WALK is wave-uniform: lanes are grouped (waterfall over ballot/readlane) by the key
// that selects their policy buckets - (scope chain start, policy version, resource kind |
// principal id) - and each group walks its scope chain, directory buckets, rule rows and CEL
// programs ONCE on uniform (scalar-unit) values; per-lane data carries the role match, the action
// masks and the condition results.  Rows of one bucket are visited in binding order exactly like
// Index.Query's result (index/index.go:214-336); role-policy synthetic DENYs first
// (index.go:318-322, 352-530).  The host flattener orders requests by that key so that a wave
// usually holds one group.
//
// Discipline (also what tests/hostsim emulates): no lane returns early; every cross-lane call
// (wave_ballot / wave_readlane / run_uniform) is reached by all 64 lanes under uniform control
// flow.  Divergent branches contain only per-lane code.
//
// Two instantiations: GENERIC = true carries the operand-stack interpreter (LDS stack, a real
// function call); GENERIC = false is selected by the host when every program of the table is a
// fused leaf or a tree of fused leaves (the common case).
#pragma once
#include "cbh_interp.h"

struct RoleSet {   // [role] ++ ancestors(role) for the request's resource scope (index.go:716-742)
  u32 role; u32 par_off; u32 par_cnt; u64 gbits;   // gbits: OR of role-dimension glob bits over the set
  u64 classes;                                      // OR of 1 << role class over the set (CBH_SEC_ROLE_CLASS)
};

// `globbit` = CBH_PAT_GLOB, or 0 in kernels for tables without glob patterns (the branch folds away)
__device__ __forceinline__ bool roleset_has(const TableDev& t, const RoleSet& rs, u32 pref, u32 globbit) {
  if (globbit && pref == CBH_PAT_ANY) return true;   // ("*" counts as a glob pattern for the kernel classes)
  if (pref & globbit) return ((rs.gbits >> (pref & 63u)) & 1ull) != 0;
  if (pref == rs.role) return true;
  for (u32 k = 0; k < rs.par_cnt; ++k) if (t.pool[rs.par_off + k] == pref) return true;
  return false;
}

// Table records are stored row-major and naturally aligned so that one wide scalar load
// (s_load_dwordx4 / x8) fetches a whole record at a wave-uniform index.
// A rule record is 16 dwords in two 8-dword halves (cbh_blob.h CbhRowField): the hot half - effect, condition
// references, role / action CLASS masks - is what every visit reads; the pattern half holds the pattern
// references and is read only for records whose masks do not decide the match (globs, class overflow) and for
// principal-policy rows.
struct __attribute__((aligned(32))) TblRow { u32 flags, cond, drcond, policy, rm_lo, rm_hi, am_lo, am_hi; };
struct LeafRec;
struct __attribute__((aligned(32))) TblRowPat { u32 action, role, resource, counts, a1, a2, r1, r2; };
struct __attribute__((aligned(16))) TblRp { u32 resource, allow_off, allow_cnt, cond; };
struct __attribute__((aligned(16))) TblDr { u32 name, parents_off, parents_cnt, cond; };
struct __attribute__((aligned(32))) TblSlot { u32 k0, k1, k2, k3, v0, v1, v2, v3; };

template <typename R, typename P>
__device__ __forceinline__ R uload_rec(P base, u32 idx) {   // P: pointer to u32 in any address space
#ifndef CBH_HOSTSIM
  static_assert(sizeof(R) == 16 || sizeof(R) == 32 || sizeof(R) == 64, "table records are 4, 8 or 16 dwords");
  const unsigned long long addr = uniform_addr((unsigned long long)(base + (size_t)idx * (sizeof(R) / 4)));
  R r;
  if constexpr (sizeof(R) == 64) {
    typedef u32 u32x16 __attribute__((ext_vector_type(16)));
    const u32x16 v = *(const __attribute__((address_space(4))) u32x16*)addr;
    __builtin_memcpy(&r, &v, 64);
  } else if constexpr (sizeof(R) == 32) {
    typedef u32 u32x8 __attribute__((ext_vector_type(8)));
    const u32x8 v = *(const __attribute__((address_space(4))) u32x8*)addr;
    __builtin_memcpy(&r, &v, 32);
  } else {
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *(const __attribute__((address_space(4))) u32x4*)addr;
    __builtin_memcpy(&r, &v, 16);
  }
  return r;
#else
  R r;   // the host copy of the image carries no alignment guarantee
  __builtin_memcpy(&r, base + (size_t)idx * (sizeof(R) / 4), sizeof(R));
  return r;
#endif
}

// directory probe with wave-uniform key -> wave-uniform result
__device__ inline bool udir_find(const TableDev& t, u32 k0, u32 k1, u32 k2, u32 k3, uint4& v) {
  u32 i = hash4(k0, k1, k2, k3) & t.hash_mask;
  for (u32 probe = 0; probe <= t.hash_mask; ++probe) {
    const TblSlot s = uload_rec<TblSlot>((const CBH_G u32*)t.hash, i);
    if (s.k0 == CBH_NONE) return false;
    if (s.k0 == k0 && s.k1 == k1 && s.k2 == k2 && s.k3 == k3) {
      v.x = s.v0; v.y = s.v1; v.z = s.v2; v.w = s.v3;
      return true;
    }
    i = (i + 1) & t.hash_mask;
  }
  return false;
}

__device__ __forceinline__ u32 uchain_next(const TableDev& t, u32 si, u32 flagbit) {   // uniform si
  while (si != CBH_NONE && !(uload(&t.scope_flags[si]) & flagbit)) si = uload(&t.scope_parent[si]);
  return si;
}

// first scope of the chain for a wave-uniform request scope word (ruletable.go:848-882)
__device__ __forceinline__ u32 uchain_first(const TableDev& t, u32 raw, u32 flagbit, bool lenient) {
  const u32 si = raw & ~CBH_SCOPE_EXACT;
  const bool exact = (raw & CBH_SCOPE_EXACT) != 0;
  if (!lenient && !(exact && (uload(&t.scope_flags[si]) & flagbit))) return CBH_NONE;
  return uchain_next(t, si, flagbit);
}

// does one of the request's roles (or an ancestor of one) appear in the derived role's parent list?
// (internal.SetIntersects(dr.ParentRoles, includingParentRoles), check.go:244)
__device__ inline bool lane_has_parent_role(const TableDev& t, const BatchDev& b, u32 poff, u32 pcnt, u32 role_off,
                                            u32 role_cnt, u32 scope_key, bool has_parents) {
  for (u32 r = 0; r < role_cnt; ++r) {
    const u32 role = b.roles[role_off + r];
    uint4 pv; u32 aoff = 0, acnt = 0;
    if (has_parents && scope_key != CBH_NONE && dir_find(t, CBH_B_PARENTS, scope_key, role, 0, pv)) { aoff = pv.x; acnt = pv.y; }
    for (u32 k = 0; k < pcnt; ++k) {
      const u32 want = t.pool[poff + k];
      if (want == role) return true;
      for (u32 a = 0; a < acnt; ++a) if (t.pool[aoff + a] == want) return true;
    }
  }
  return false;
}

// One fused leaf (OP_LEAF_BIN word + two operand words) for a lane: 0 false, 1 true, 3 CEL error.
struct __attribute__((aligned(16))) TblVal { u32 tag, pad, lo, hi; };
template <typename P>
__device__ __forceinline__ Val uval(P recs, u32 idx) {   // constant / constant-heap entry, scalar load
  const TblVal r = uload_rec<TblVal>(recs, idx);
  return mk(r.tag, (u64)r.lo | ((u64)r.hi << 32));
}

__device__ __forceinline__ int leaf_value(const Ctx& c, Lane& L, u32 w, u32 a0, u32 a1) {
  const u32 a = w >> 8;
  const u32 ka = (a >> 8) & 0xF, kb = (a >> 12) & 0xF, op = a & 0xFF;   // all wave-uniform
  const Val x = ka == 0 ? uval(c.t.const_rec, a0) : load_operand(c, L, ka, a0);
  const Val y = kb == 0 ? uval(c.t.const_rec, a1) : load_operand(c, L, kb, a1);
  if (op == OP_IN && kb == 0 && y.t == CBH_T_LIST && cont_sel(y.v) == CBH_HEAP_TABLE) {
    // membership in a constant list: the elements are uniform, walk them on the scalar unit
    if (x.t == CBH_T_ERR) return 3;
    const u32 n = cont_len(y.v), off = cont_off(y.v);
    int found = 0; bool slow = false;
    for (u32 i = 0; i < n; ++i) {
      const int e = fast_equal(x, uval(c.t.theap_rec, off + i));
      if (e == -2) slow = true; else found |= e;
    }
    if (!slow) return found;
  }
  const int f = fast_compare(c, op, x, y);
  if (f >= 0) return f;
  if (f == -1) return 3;
  const SlowVal v = compare_op_slow(c.ka_mem, L.req, a & 0xFF, x, y);
  L.status |= v.status;
  if (v.t == CBH_T_ERR) return 3;
  return (v.t == CBH_T_BOOL && v.v) ? 1 : 0;
}

// all/any/none tree of fused leaves, out of line: a real call keeps the (hot) single-leaf path and the
// rule-row loop around it small.  Works from the launch arguments in memory like run_uniform.
#ifndef CBH_HOSTSIM
__attribute__((noinline))
#endif
__device__ u32 eval_leaf_tree(const KernelArgs* ka, const VmLds lds, u32 req, u32 pc, bool active) {
  const Ctx c = ctx_from_memory(uniform_ptr(ka), lds);
  Lane L; L.req = req; L.edr = 0; L.status = 0; L.edr_err = false;   // leaves never read runtime.*
  const bool strict = (c.flags & CBH_F_STRICT_EVALUATION) != 0;
  pc = uniform(pc);
    bool live = active, last = false;
    int result = 0;
    u32 saved = 0, acc = 0, depth = 0;
    for (;;) {
      const u32 w = uload(&c.t.code[pc]); ++pc;
      const u32 op = w & 0xFFu, a = w >> 8;
      if (op == OP_LEAF_BIN) {
        const u32 a0 = uload(&c.t.code[pc]), a1 = uload(&c.t.code[pc + 1]); pc += 2;
        last = false;
        if (live) {   // leaves after the deciding one are not evaluated (check.go:697-749)
          const int v = leaf_value(c, L, w, a0, a1);
          if (v == 3) { L.status |= CBH_ST_CEL_ERROR; if (strict) { result = 2; live = false; } }
          last = v == 1;
        }
      } else if (op == OP_TREE_BEGIN) {
        const u32 bit = 1u << depth;
        saved = live ? (saved | bit) : (saved & ~bit);
        acc = (a == 0) ? (acc | bit) : (acc & ~bit);
        ++depth;
      } else if (op == OP_TREE_ACC) {
        const u32 bit = 1u << (depth - 1);
        if (live) {
          if (a == 0) { if (!last) { acc &= ~bit; live = false; } }
          else if (last) { acc |= bit; live = false; }
        }
      } else if (op == OP_TREE_END) {
        --depth;
        const u32 bit = 1u << depth;
        if (result != 2) live = (saved & bit) != 0;
        last = ((acc & bit) != 0) != (a == 2);
      } else break;   // OP_RET
    }
    const u32 r = result == 2 ? 2u : ((active && last) ? 1u : 0u);
    return r | (L.status << 8);   // result | status bits raised while evaluating
}

// Evaluate a condition reference for the lanes with active=true (all lanes call together).
// Per lane: 0 = not satisfied, 1 = satisfied, 2 = strict-mode evaluation error.
//   CBH_COND_LEAF     one fused leaf: evaluated inline
//   CBH_COND_LEAFTREE all/any/none tree whose leaves are all fused leaves: inline, no operand
//                     stack (each TREE_ACC consumes the value its child just produced)
//   otherwise         the operand-stack interpreter (GENERIC instantiation only)
// A real call on purpose: the table walk around it then carries none of the evaluator's code or
// registers, and the three places that evaluate conditions (derived roles, role policies, rule rows)
// share one copy.  Returns result | status bits << 8.
#ifndef CBH_HOSTSIM
__attribute__((noinline))
#endif
__device__ u32 eval_leaf_one(const KernelArgs* ka, const VmLds lds, u32 req, u32 ref, bool active) {   // CBH_COND_LEAF
  const Ctx c = ctx_from_memory(uniform_ptr(ka), lds);
  const u32 pc = ref & CBH_COND_PC_MASK;
  const u32 w = uload(&c.t.code[pc]), a0 = uload(&c.t.code[pc + 1]), a1 = uload(&c.t.code[pc + 2]);
  Lane L; L.req = req; L.edr = 0; L.status = 0; L.edr_err = false;   // leaves never read runtime.*
  u32 r = 0;
  if (active) {
    r = (u32)leaf_value(c, L, w, a0, a1);
    if (r == 3) { L.status |= CBH_ST_CEL_ERROR; r = (c.flags & CBH_F_STRICT_EVALUATION) ? 2u : 0u; }
  }
  return r | (L.status << 8);
}
// Every real call is made from the caller's own frame and none of the callees calls on (a function between the caller and
// run_uniform would add the registers it keeps across that call to run_uniform's, and the sum - not the larger - is what the kernel
// is sized by: 257 registers, one wave to a SIMD, where 253 fit two).
template <bool GENERIC>
__device__ __forceinline__ u32 eval_ref(const KernelArgs* ka, const VmLds lds, u32 req, u64 edr, bool edr_err, u32 ref, bool active) {
  ref = uniform(ref);
  if (ref & CBH_COND_LEAF) return eval_leaf_one(ka, lds, req, ref, active);
  if (ref & CBH_COND_LEAFTREE) return eval_leaf_tree(ka, lds, req, ref & CBH_COND_PC_MASK, active);
  if (GENERIC) return run_uniform(ka, lds, req, edr, edr_err, ref, active);
  return active ? ((u32)CBH_ST_UNSUPPORTED << 8) : 0u;   // unreachable: the host picks the GENERIC kernel for such tables
}

// A CBH_COND_LEAF program is an 8-dword record on an 8-dword boundary of the tape (celc.py
// condition_program): instruction, operands and the value of its constant operand in one scalar load.
struct __attribute__((aligned(32))) LeafRec { u32 w, a0, a1, ret, ctag, clo, chi, pad; };
struct __attribute__((aligned(64))) TblRowFull { TblRow hot; LeafRec leaf; };   // a whole 16-dword rule record

// The common outcome of a fused leaf, inline in the table walk: both operands present, same-typed
// scalars (or plainly unequal types), or an operand missing.  Returns 0 / 1, 3 = CEL error, or 4 when
// this lane needs the full evaluator (mixed numeric types, containers, lists that are not table constants).
__device__ __forceinline__ u32 leaf_fast(const Ctx& c, const Lane& L, const LeafRec& lr) {
  const u32 a = lr.w >> 8;
  const u32 ka = (a >> 8) & 0xF, kb = (a >> 12) & 0xF, op = a & 0xFF;   // all wave-uniform
  // The shapes conditions usually have are classified by the lowering (celc.py leaf_class) and run
  // as straight-line code behind ONE uniform branch; each arm answers only what it is sure of and
  // sends the rest (mixed numeric types, orderings of mismatched types) to the full evaluator.
  switch (lr.pad) {
    case 1: {   // cached column ==/!= string or bool constant
      const Val x = cached_column(c, L, lr.a0);
      if (x.t == CBH_T_ERR) return 3;
      const bool eq = x.t == lr.ctag && (u32)x.v == lr.clo;   // other types are plainly unequal
      return (u32)(eq == (op == OP_EQ));
    }
    case 2: {   // cached column <op> double constant
      const Val x = cached_column(c, L, lr.a0);
      if (x.t == CBH_T_ERR) return 3;
      if (x.t != CBH_T_DOUBLE) return (x.t == CBH_T_INT || x.t == CBH_T_UINT) ? 4u : op == OP_EQ ? 0u : op == OP_NE ? 1u : 4u;
      const double p = as_f64(x.v), q = as_f64((u64)lr.clo | ((u64)lr.chi << 32));
      return (op == OP_EQ) ? p == q : (op == OP_NE) ? p != q : (op == OP_LT) ? p < q : (op == OP_LE) ? p <= q
           : (op == OP_GT) ? p > q : p >= q;   // NaN: every ordering false, != true
    }
    case 3: {   // cached column ==/!= cached column
      const Val x = cached_column(c, L, lr.a0), y = cached_column(c, L, lr.a1);
      if (x.t == CBH_T_ERR || y.t == CBH_T_ERR) return 3;
      const int e = fast_equal(x, y);
      return e < 0 ? 4u : (u32)(op == OP_EQ ? e : 1 - e);
    }
    case 4: {   // cached column ==/!= P.id (either order)
      const Val x = cached_column(c, L, ka == 3 ? lr.a0 : lr.a1);
      if (x.t == CBH_T_ERR) return 3;
      const bool eq = x.t == CBH_T_STRING && (u32)x.v == L.pid;
      return (u32)(eq == (op == OP_EQ));
    }
    case 5: {   // cached column in [string constants]
      const Val x = cached_column(c, L, lr.a0);
      if (x.t == CBH_T_ERR) return 3;
      const u32 n = lr.clo, off = lr.chi & 0x3FFFFFFFu;   // list payload: sel:2 | off:30 | len:32
      u32 found = 0;
      for (u32 i = 0; i < n; ++i) found |= (u32)((u32)x.v == uload(&c.t.theap_rec[4 * (size_t)(off + i) + 2]));
      return x.t == CBH_T_STRING ? found : 0u;   // a non-string equals no string
    }
    case 6: {   // cached column in [at most three string constants]: their ids are in the record (CBH_NONE pads)
      const Val x = cached_column(c, L, lr.a0);
      if (x.t == CBH_T_ERR) return 3;
      const u32 v = (u32)x.v;
      return x.t == CBH_T_STRING ? (u32)((v == lr.ctag) | (v == lr.clo) | (v == lr.chi)) : 0u;   // a non-string equals no string
    }
    default: break;
  }
  if (lr.ctag == CBH_NONE && (ka == 0 || kb == 0)) return 4;
  const Val cv = mk(lr.ctag, (u64)lr.clo | ((u64)lr.chi << 32));
  // the lowering tells the two cheap operand kinds apart (celc.py): 3 = a column parked in LDS,
  // 4 = P.id, which is already in a register; the rest goes through the general loader
  // (operands that need a memory access - kinds 1 and 2 - are left to the full evaluator)
  if (ka == 1 || ka == 2 || kb == 1 || kb == 2) return 4;
  const Val x = ka == 3 ? cached_column(c, L, lr.a0) : ka == 0 ? cv : mk(CBH_T_STRING, L.pid);
  const Val y = kb == 0 ? cv : kb == 3 ? cached_column(c, L, lr.a1) : mk(CBH_T_STRING, L.pid);
  if (x.t == CBH_T_ERR || y.t == CBH_T_ERR) return 3;   // a missing attribute: every comparison of an error is that error
  if (op == OP_EQ || op == OP_NE) {
    const int e = fast_equal(x, y);
    return e < 0 ? 4u : (u32)(op == OP_EQ ? e : 1 - e);
  }
  if (op == OP_IN) {
    if (kb != 0 || y.t != CBH_T_LIST || cont_sel(y.v) != CBH_HEAP_TABLE) return 4;   // uniform
    const u32 n = cont_len(y.v), off = cont_off(y.v);
    u32 found = 0; bool slow = false;
    for (u32 i = 0; i < n; ++i) {   // the elements are uniform: scalar loads
      const int e = fast_equal(x, uval(c.t.theap_rec, off + i));
      if (e == -2) slow = true; else found |= (u32)e;
    }
    return slow ? 4u : found;
  }
  if (x.t == CBH_T_DOUBLE && y.t == CBH_T_DOUBLE) {
    const double p = as_f64(x.v), q = as_f64(y.v);
    return (op == OP_LT) ? p < q : (op == OP_LE) ? p <= q : (op == OP_GT) ? p > q : p >= q;   // NaN: all false
  }
  if (x.t == CBH_T_INT && y.t == CBH_T_INT) {
    const i64 p = (i64)x.v, q = (i64)y.v;
    return (op == OP_LT) ? p < q : (op == OP_LE) ? p <= q : (op == OP_GT) ? p > q : p >= q;
  }
  return 4;
}

// `lr` = the fused-leaf record of `ref` when the caller already holds it (the copy embedded in a rule record)
template <bool GENERIC>
__device__ __forceinline__ int eval_cond_rec(const Ctx& c, Lane& L, u32 ref, const LeafRec& lr, bool active) {
  u32 fast = 0;
  if (active) {
    fast = leaf_fast(c, L, lr);
    if (fast == 3) {   // CEL error: the leaf counts as false, or as a DENY in strict mode (check.go:697-749)
      L.status |= CBH_ST_CEL_ERROR;
      fast = (c.flags & CBH_F_STRICT_EVALUATION) ? 2u : 0u;
    }
  }
  const bool rest = active && fast == 4;
  if (wave_ballot(rest) == 0) return (int)fast;   // the whole wave was served inline
  const u32 r = eval_ref<GENERIC>(c.ka_mem, lds_of(c), L.req, L.edr, L.edr_err, ref, rest);
  if (rest) { L.status |= r >> 8; fast = r & 0xFF; }
  return (int)fast;
}

template <bool GENERIC>
__device__ __forceinline__ int eval_cond(const Ctx& c, Lane& L, u32 ref, bool active) {
  if (ref & CBH_COND_LEAF) {
    const LeafRec lr = uload_rec<LeafRec>(c.t.code, (ref & CBH_COND_PC_MASK) >> 3);
    return eval_cond_rec<GENERIC>(c, L, ref, lr, active);
  }
  const u32 r = eval_ref<GENERIC>(c.ka_mem, lds_of(c), L.req, L.edr, L.edr_err, ref, active);
  u32 fast = 0;
  if (active) { L.status |= r >> 8; fast = r & 0xFF; }
  return (int)fast;
}

// Copy the launch arguments into registers once, with scalar loads.  Read through the pointer they
// would be re-fetched from memory (vector loads + a full wait) at every use inside the loops, because
// the compiler cannot prove the kernel's own stores leave them untouched.
__device__ __forceinline__ void load_args(KernelArgs& dst, const KernelArgs* src) {
#ifndef CBH_HOSTSIM
  typedef u32 u32x4 __attribute__((ext_vector_type(4)));
  static_assert(sizeof(KernelArgs) % 16 == 0, "KernelArgs is copied in 16-byte scalar loads");
  constexpr int N = sizeof(KernelArgs) / 16;
  const unsigned long long base = uniform_addr((unsigned long long)src);
  u32x4 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = *(const __attribute__((address_space(4))) u32x4*)(base + 16ull * i);
  __builtin_memcpy(&dst, v, sizeof(KernelArgs));
#else
  dst = *src;
#endif
}

// Column cache: issue every load of this lane's request attributes now, park them in LDS
// (async global->LDS copies, `global_load_lds_dword`: no staging registers, the loads of every
// column are in flight together; the destination of such a copy is wave-uniform base + lane * 4,
// which is exactly a [column][lane] dword plane)
__device__ __forceinline__ void fill_column_cache(const Ctx& c, const BatchDev& b, u32 NR, u32 req) {
  const bool packed = (c.flags & CBH_FI_PACKED_TAGS) != 0;   // the two forms of the tags: cbh_vm.h CBH_CC_DWORDS
  if (packed) {
    // the tag bytes of ALL columns first, back to back, then their LDS stores: a load followed by its store inside the loop
    // over the columns made that loop one round trip to memory per column (cc_load_tags below tells the story)
    CBH_L u32* tags = c.cc + 2u * c.n_cached * CBH_BLOCK;
    const u32 n = c.n_cached, last = n ? n - 1u : 0u;
    const CBH_G u8* base = n ? b.col_tag : (const CBH_G u8*)b.req_u32;
    u32 by[CBH_CACHE_COLS];
#pragma unroll
    for (u32 k = 0; k < CBH_CACHE_COLS; ++k) by[k] = (u32)base[(size_t)(k < n ? k : last) * NR + (n ? req : 0u)];
#pragma unroll
    for (u32 g = 0; g < CBH_CACHE_COLS / 4; ++g)
      if (4u * g < n) tags[g * CBH_BLOCK + c.tid] = by[4 * g] | (by[4 * g + 1] << 8) | (by[4 * g + 2] << 16) | (by[4 * g + 3] << 24);
  }
  for (u32 k = 0; k < c.n_cached; ++k) {
    const size_t ix = (size_t)k * NR + req;
    const CBH_G u32* vsrc = (const CBH_G u32*)(b.col_val + ix);
#ifndef CBH_HOSTSIM
    __builtin_amdgcn_global_load_lds((const CBH_G void*)vsrc, (CBH_L void*)(c.cc + k * CBH_BLOCK), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((const CBH_G void*)(vsrc + 1), (CBH_L void*)(c.cc + (c.n_cached + k) * CBH_BLOCK), 4, 0, 0);
    if (!packed) __builtin_amdgcn_global_load_lds((const CBH_G void*)(b.col_tag + (ix & ~(size_t)3)), (CBH_L void*)(c.cc + (2 * c.n_cached + k) * CBH_BLOCK), 4, 0, 0);
#else
    c.cc[k * CBH_BLOCK + c.tid] = vsrc[0];
    c.cc[(c.n_cached + k) * CBH_BLOCK + c.tid] = vsrc[1];
    // (the host arrays carry no slack after their last byte: place the one byte instead of copying its dword)
    if (!packed) c.cc[(2 * c.n_cached + k) * CBH_BLOCK + c.tid] = (u32)b.col_tag[ix] << ((ix & 3u) * 8u);
#endif
  }
}

// one dword per lane, global memory -> LDS at `lds` + lane * 4 (`lds` wave-uniform).  On the device an asynchronous copy that needs no
// staging register; the simulator performs the same copy at the same addresses, so that the address arithmetic of the callers runs
// in the CPU tier too.  (The instruction's own offset field is deliberately not offered: it moves BOTH sides of the copy.)
__device__ __forceinline__ void lds_dma_dword(const CBH_G void* g, CBH_L u32* lds, u32 lane) {
#ifndef CBH_HOSTSIM
  (void)lane;
  __builtin_amdgcn_global_load_lds(g, (CBH_L void*)lds, 4, 0, 0);
#else
  __builtin_memcpy(&lds[lane], g, 4);
#endif
}
// The same fill for the kernels whose lanes hold CONSECUTIVE requests (the flat kernels, the walk), in TWO steps the caller places:
// every address is a wave-uniform base - the column's plane at the wave's first request, formed on the scalar unit - plus a 32-bit
// lane offset: no 64-bit address arithmetic per lane and column, and the LDS side (the wave's slice of the cache: uniform) goes to
// M0 without a read-back from a vector register.  `req0` = the wave's first request (uniform), `d` = this lane's distance from it
// (< 64; a lane beyond the batch's end passes 0: it shadows the wave's first request here, and never stores).
//
// Why two steps.  In the packed form of the tags a lane's tag byte goes through a register (there is no one-byte copy into LDS),
// and a load followed by its LDS store INSIDE the loop over the columns made that loop a chain of round trips to memory - one per
// column, nine for the tables of C3 / C4 / T (the store waits for its byte, and with it for every copy issued before).  So:
//   cc_load_tags   the tag bytes of ALL columns into registers, back to back, depending on nothing - issued with the request's
//                  other first loads;
//   cc_fill        the bytes into LDS (packed form), then every column's asynchronous copies.  Any LDS access the compiler
//                  cannot tell apart from the copies' destinations waits for all of them, so the caller issues the copies as
//                  late as it can: behind its own LDS stores, beside its last dependent loads.
struct CcTags { u32 w[CBH_CACHE_COLS / 4]; };
__device__ __forceinline__ CcTags cc_load_tags(const Ctx& c, const BatchDev& b, u32 NR, u32 req0, u32 d) {
  CcTags t;
#pragma unroll
  for (u32 g = 0; g < CBH_CACHE_COLS / 4; ++g) t.w[g] = 0;
  if ((c.flags & CBH_FI_PACKED_TAGS) == 0) return t;   // (uniform; the wide form's tags arrive as copies of their dwords)
  const u32 n = c.n_cached, last = n ? n - 1u : 0u;
  // (no branch around a load: a load inside a conditional block is waited for at the block's end.  A column beyond the last
  // re-reads the last one's byte - the same line - and a table without columns reads a request word instead)
  const CBH_G u8* base = n ? b.col_tag : (const CBH_G u8*)b.req_u32;
  u32 by[CBH_CACHE_COLS];
#pragma unroll
  for (u32 k = 0; k < CBH_CACHE_COLS; ++k) {
    const u32 kk = k < n ? k : last;
    const size_t u = (size_t)kk * NR + req0;   // uniform
    by[k] = (u32)(base + u)[d];
  }
#pragma unroll
  for (u32 g = 0; g < CBH_CACHE_COLS / 4; ++g) t.w[g] = by[4 * g] | (by[4 * g + 1] << 8) | (by[4 * g + 2] << 16) | (by[4 * g + 3] << 24);
  return t;
}
__device__ __forceinline__ void cc_fill(const Ctx& c, const BatchDev& b, u32 NR, u32 req0, u32 d, const CcTags& t) {
  const bool packed = (c.flags & CBH_FI_PACKED_TAGS) != 0;
  CBH_L u32* tags = c.cc + 2u * c.n_cached * CBH_BLOCK;
  if (packed) {
#pragma unroll
    for (u32 g = 0; g < CBH_CACHE_COLS / 4; ++g) if (4u * g < c.n_cached) tags[g * CBH_BLOCK + c.tid] = t.w[g];
  }
  const u32 d8 = d * 8u;
  for (u32 k = 0; k < c.n_cached; ++k) {
    const size_t u = (size_t)k * NR + req0;   // uniform
    const CBH_G char* vb = (const CBH_G char*)(b.col_val + u);
    const CBH_G char* vb4 = vb + 4;   // (the value's high word from a second uniform base)
    lds_dma_dword(vb + d8, c.cc + k * CBH_BLOCK, c.tid);
    lds_dma_dword(vb4 + d8, c.cc + (c.n_cached + k) * CBH_BLOCK, c.tid);
    if (!packed) {   // the aligned dword that holds the lane's tag byte (cached_tag picks the byte)
      const u32 sh = (u32)u & 3u;
#ifndef CBH_HOSTSIM
      const CBH_G char* tb = (const CBH_G char*)(b.col_tag + (u - sh));
      lds_dma_dword(tb + ((d + sh) & ~3u), c.cc + (2 * c.n_cached + k) * CBH_BLOCK, c.tid);
#else
      // (the host arrays carry no slack after their last byte and need not be dword-aligned: place the one byte where the device's
      // aligned dword would have it)
      c.cc[(2 * c.n_cached + k) * CBH_BLOCK + c.tid] = (u32)b.col_tag[u + d] << (((d + sh) & 3u) * 8u);
#endif
    }
  }
}

struct CbhPassPrincipal { static constexpr bool value = false; };   // tags of the two instantiations of the
struct CbhPassResource { static constexpr bool value = true; };     // policy pass (check_body below)

#define CBH_BUCKET_MEMO 8u          /* chain positions whose directory answer is kept across a group's role iterations */
#define CBH_FEAT_DERIVED_ROLES 1   /* FEAT bits: what the table uses, compiled in only then */
#define CBH_FEAT_ROLE_POLICIES 2  /* role policies and / or parent roles */
#define CBH_FEAT_GLOBS 4          /* glob patterns in some dimension (action / role / kind) */
#define CBH_FEAT_PRINCIPAL_POLICIES 8
#define CBH_FEAT_ALL 15
#define CBH_FEAT_MAX4 16          /* a property of the batch, not the table: at most four actions per request */
#define CBH_FEAT_TRACE 32         /* the trace pass (cbh_trace_batch): conditions run as trace programs, errors and outputs are logged */
#define CBH_FEAT_TRAIL 64         /* cbh_check_batch_trail: the policy of every binding iterated is marked (its own kernel: the marks cost the */
                                  /* general walk the registers that let it run two waves to a SIMD) */
struct __attribute__((aligned(32))) TblTraceRow { u32 cond, drcond, vars_off, vars_cnt, drvars_off, drvars_cnt, out_activated, out_not_met; };
struct __attribute__((aligned(16))) TblTraceCond { u32 cond, vars_off, vars_cnt, pad; };
struct __attribute__((aligned(32))) TblTraceRp { u32 cond, vars_off, vars_cnt, out_activated, out_not_met, pad0, pad1, pad2; };
#define CBH_TR_DRFAIL 32u         /* w1 bit 5 of an output record: the rule's derived-role condition was not satisfied (the host */
                                  /* drops the first such visit per evaluation key, as check.go:343-347 emits nothing there)    */

template <bool GENERIC, typename AM, int FEAT>   // AM: per-request action mask, u32 when no request of the batch carries more than 32 actions
__device__ __forceinline__ void check_body(const KernelArgs& ka_regs, Ctx& c) {
  const TableDev& t = ka_regs.t;
  const BatchDev& b = ka_regs.b;
  const OutDev& o = ka_regs.o;
  const u32 flags = ka_regs.flags;

#ifdef CBH_PROFILE_CYCLES   // profiling build only (tools/gpu_cycles.py): the counters cost ~20 SGPRs
  const u64 cyc_start = __builtin_readcyclecounter();
#endif
  const u32 rix = b.req_lo + blockIdx.x * CBH_BLOCK + threadIdx.x;
  const u32 NR = b.n_requests;
  bool valid = rix < b.req_hi;
  if ((flags & (CBH_FI_ONLY_WIDE | CBH_FI_ONLY_WIDER)) && valid) {   // the requests the cbh_walk2 kernels decide are not this launch's (cbh_vm.h CBH_FI_*)
    const u32 na = b.req_u32[(size_t)CBH_RQ_ACT_CNT * NR + rix], nr = b.req_u32[(size_t)CBH_RQ_ROLE_CNT * NR + rix];
    valid = (flags & CBH_FI_ONLY_WIDER) ? cbh_is_wider(na, nr) : cbh_is_wide(na, nr);
  }
  const u32 req = valid ? rix : b.req_lo;   // tail lanes shadow the chunk's first request and never store
#define RQ(f) b.req_u32[(size_t)(f) * NR + req]
  const u32 pid = RQ(CBH_RQ_PRINCIPAL_ID);
  const u32 p_scope = RQ(CBH_RQ_P_SCOPE), p_ver = RQ(CBH_RQ_P_VERSION);
  const u32 kind = RQ(CBH_RQ_KIND);
  const u32 r_scope = RQ(CBH_RQ_R_SCOPE), r_ver = RQ(CBH_RQ_R_VERSION);
  const u32 role_off = RQ(CBH_RQ_ROLE_OFF), role_cnt = RQ(CBH_RQ_ROLE_CNT);
  const u32 act_off = RQ(CBH_RQ_ACT_OFF);
  const u32 act_cnt = valid ? RQ(CBH_RQ_ACT_CNT) : 0;   // <= 64 (the flattener splits larger requests)
#undef RQ
  fill_column_cache(c, b, NR, req);
  constexpr u32 AM_BITS = sizeof(AM) * 8;
  const AM all = act_cnt >= AM_BITS ? (AM)~(AM)0 : (AM)(((AM)1 << act_cnt) - 1);
  // the first four action ids stay in registers (requests rarely carry more)
  // (unconditional loads - an action that does not exist reads element 0 and is masked - so that they all go
  // out together instead of one guarded round trip after another)
  const u32 l0 = b.tuple_action[act_cnt > 0 ? act_off : 0u], l1 = b.tuple_action[act_cnt > 1 ? act_off + 1 : 0u];
  const u32 l2 = b.tuple_action[act_cnt > 2 ? act_off + 2 : 0u], l3 = b.tuple_action[act_cnt > 3 ? act_off + 3 : 0u];
  const u32 a0 = act_cnt > 0 ? l0 : CBH_NONE, a1 = act_cnt > 1 ? l1 : CBH_NONE, a2 = act_cnt > 2 ? l2 : CBH_NONE, a3 = act_cnt > 3 ? l3 : CBH_NONE;
  // their action classes (CBH_SEC_ACTION_CLASS; 63 = not a literal rule action), for the records whose class
  // masks decide the match
  u32 ac0 = 63u, ac1 = 63u, ac2 = 63u, ac3 = 63u;
  if ((FEAT & CBH_FEAT_MAX4) != 0) {
    const u32 kmax = t.K ? t.K - 1u : 0u;
    const u32 c0 = t.action_class[a0 < t.K ? a0 : kmax], c1 = t.action_class[a1 < t.K ? a1 : kmax];
    const u32 c2 = t.action_class[a2 < t.K ? a2 : kmax], c3 = t.action_class[a3 < t.K ? a3 : kmax];
    ac0 = a0 < t.K ? c0 : 63u; ac1 = a1 < t.K ? c1 : 63u; ac2 = a2 < t.K ? c2 : 63u; ac3 = a3 < t.K ? c3 : 63u;
  }
  // ... and the first two roles: fetched with the rest of the request instead of one memory round
  // trip at the head of every role iteration
  const u32 lr0 = b.roles[role_cnt > 0 ? role_off : 0u], lr1 = b.roles[role_cnt > 1 ? role_off + 1 : 0u];
  const u32 role0 = role_cnt > 0 ? lr0 : 0, role1 = role_cnt > 1 ? lr1 : 0;

  const bool lenient = (flags & CBH_F_LENIENT_SCOPE_SEARCH) != 0;
  const bool strict = (flags & CBH_F_STRICT_EVALUATION) != 0;
  constexpr bool F_DR = (FEAT & CBH_FEAT_DERIVED_ROLES) != 0, F_RP = (FEAT & CBH_FEAT_ROLE_POLICIES) != 0;
  constexpr bool F_GLOB = (FEAT & CBH_FEAT_GLOBS) != 0;
  constexpr bool F_PP = (FEAT & CBH_FEAT_PRINCIPAL_POLICIES) != 0;
  constexpr bool F_MAX4 = (FEAT & CBH_FEAT_MAX4) != 0;   // no request of the batch has more than four actions
  constexpr bool TRACE = (FEAT & CBH_FEAT_TRACE) != 0;
  constexpr u32 GLOBBIT = F_GLOB ? CBH_PAT_GLOB : 0u;   // no glob patterns in the table: every pattern reference is a literal
  auto pmatch = [&](u32 pref, u32 sid, u64 bits) -> bool { return (F_GLOB && pref == CBH_PAT_ANY) || ((pref & GLOBBIT) ? ((bits >> (pref & 63u)) & 1ull) != 0 : pref == sid); };
  const bool want_edr = F_DR && ((flags & CBH_F_WANT_DERIVED_ROLES) != 0 || (t.flags & CBH_MF_USES_RUNTIME_EDR) != 0);
  const bool has_parents = F_RP && (t.flags & CBH_MF_HAS_PARENT_ROLES) != 0;
  const bool has_rolepol = F_RP && (t.flags & CBH_MF_HAS_ROLE_POLICIES) != 0;
  const bool want_ps = o.policy != nullptr || o.scope != nullptr;
  const bool want_ep = (FEAT & CBH_FEAT_TRAIL) != 0 && (flags & CBH_F_WANT_EFFECTIVE_POLICIES) != 0 && o.eff_pol != nullptr;   // cbh_check_batch_trail
  u32 ep_last = CBH_NONE;   // the policy this lane marked last (a bucket's records are one policy's: one mark per bucket and role, not per record)
  auto ep_note = [&](u32 policy) { if (policy != ep_last) { ep_mark(o, b, req, policy); ep_last = policy; } };

  Lane L; L.req = req; L.edr = 0; L.status = 0; L.edr_err = false; L.pid = pid; L.edr_errmask = 0;
  u64 edr_acc = 0;

  // mask of this request's actions matching an action-dimension pattern reference
  auto match_actions = [&](u32 pat) -> AM {
    AM m = 0;
    if (F_GLOB && pat == CBH_PAT_ANY) return all;
    if (!(pat & GLOBBIT)) {
      m = (AM)(a0 == pat) | ((AM)(a1 == pat) << 1) | ((AM)(a2 == pat) << 2) | ((AM)(a3 == pat) << 3);
      if (!F_MAX4) for (u32 k = 4; k < act_cnt; ++k) m |= (AM)(b.tuple_action[act_off + k] == pat) << k;
    } else {
      const u32 gi = pat & 63u;
      for (u32 k = 0; k < act_cnt; ++k)
        m |= (AM)((gbits_of(t, b, DIM_ACTION, b.tuple_action[act_off + k]) >> gi) & 1ull) << k;
    }
    return m & all;
  };
  // policy / scope outputs are written at the moment an action's (tentative) result changes
  // The first four actions park theirs in LDS until the end (no global stores in the middle of the
  // kernel: vmcnt is in-order, an early store would sit in front of every later load's wait; and
  // not in registers: eight rarely touched VGPRs are what pushes the kernel over its 128-VGPR budget).
  __shared__ u32 ps_lds[8 * CBH_BLOCK];
  __shared__ AM drm_lds[3 * CBH_BLOCK];   // derived-role outcome memo, see the walk below
  __shared__ u32 bucket_memo_lds[CBH_BUCKET_MEMO * 5];   // directory answers per chain position, see the role loop
  CBH_L u32* bucket_memo = (CBH_L u32*)bucket_memo_lds;
#define DRM(i) drm_lds[(i) * CBH_BLOCK + c.tid]
#define PS_POL(k) ps_lds[(k) * CBH_BLOCK + c.tid]
#define PS_SCP(k) ps_lds[(4 + (k)) * CBH_BLOCK + c.tid]
  auto write_ps = [&](AM mask, u32 polw, u32 scpw) {
    if (!want_ps) return;
    if (mask & 1) { PS_POL(0) = polw; PS_SCP(0) = scpw; }
    if (mask & 2) { PS_POL(1) = polw; PS_SCP(1) = scpw; }
    if (mask & 4) { PS_POL(2) = polw; PS_SCP(2) = scpw; }
    if (mask & 8) { PS_POL(3) = polw; PS_SCP(3) = scpw; }
    mask &= F_MAX4 ? (AM)0 : ~(AM)0xF;
    while (mask) {
      const u32 k = (u32)__builtin_ctzll((u64)mask);
      mask &= mask - 1;
      if (o.policy) o.policy[act_off + k] = polw;
      if (o.scope) o.scope[act_off + k] = scpw;
    }
  };

  // kind-dimension glob bits of the resource kind: looked up where a pattern needs them (zero loads
  // for a table without kind globs) rather than held in two registers for the whole kernel
#define KIND_BITS() (F_GLOB ? gbits_of(t, b, DIM_KIND, kind) : 0ull)
  // parent roles are looked up with the request's own resource scope only (check.go:172,227)
  const u32 pr_scope_key = (r_scope & CBH_SCOPE_EXACT) ? (r_scope & ~CBH_SCOPE_EXACT) : CBH_NONE;

  // does a rule record's role list (wave-uniform) match this lane's [role] ++ ancestors?  Decided by the class
  // mask where the lowering says it is exact, else by the pattern references of the record's second half.
  auto pat_role_match = [&](const TblRow& rw, const TblRowPat& pt, const RoleSet& rs, bool active) -> bool {
    if (rw.flags & CBH_ROW_F_ROLE_BY_CLASS) return active && ((((u64)rw.rm_lo | ((u64)rw.rm_hi << 32)) & rs.classes) != 0);
    const u32 n_role = pt.counts >> 16;   // 0 = a single inline reference
    bool m = false;
    if (rw.flags & CBH_ROW_F_ROLE_LIST) {   // more than three roles: the list lives in the pool
      for (u32 i = 0; i < n_role; ++i) { const u32 pref = uload(&t.pool[pt.role + i]); m = m || (active && roleset_has(t, rs, pref, GLOBBIT)); }
    } else if (active) {
      m = roleset_has(t, rs, pt.role, GLOBBIT);
      if (n_role > 1) m = m || roleset_has(t, rs, pt.r1, GLOBBIT);
      if (n_role > 2) m = m || roleset_has(t, rs, pt.r2, GLOBBIT);
    }
    return m;
  };
  auto rec_role_match = [&](const TblRow& rw, u32 row, const RoleSet& rs, bool active) -> bool {
    if (rw.flags & CBH_ROW_F_ROLE_BY_CLASS) return active && ((((u64)rw.rm_lo | ((u64)rw.rm_hi << 32)) & rs.classes) != 0);
    const TblRowPat pt = uload_rec<TblRowPat>(t.rowpat, row);
    return pat_role_match(rw, pt, rs, active);
  };

  // ---- routing preamble: scope chains and existence (check.go:116-121, 165-170), resolved once per
  // distinct route of the wave on the scalar unit (a sorted batch has one route per wave)
  u32 p_first = CBH_NONE, r_first = CBH_NONE;
  bool p_exists = false, r_exists = false;
  {
    bool pendq = true;
    for (;;) {
      const u64 remq = wave_ballot(pendq);
      if (remq == 0) break;
      const u32 lead = first_lane(remq);
      const u32 g_ps = wave_readlane(p_scope, lead), g_pv = wave_readlane(p_ver, lead), g_rs = wave_readlane(r_scope, lead),
                g_rv = wave_readlane(r_ver, lead), g_k = wave_readlane(kind, lead);
      const u64 g_kbits = wave_readlane64(KIND_BITS(), lead);
      // (a table without principal policies has no scope in the principal map: its kernels leave the
      // principal side out altogether)
      const bool inq = pendq && (!F_PP || (p_scope == g_ps && p_ver == g_pv)) && r_scope == g_rs && r_ver == g_rv && kind == g_k;
      pendq = pendq && !inq;
      const u32 pf = F_PP ? uchain_first(t, g_ps, FLAG_PRIN, lenient) : CBH_NONE, rf = uchain_first(t, g_rs, FLAG_RES, lenient);
      bool pe = false, re = false;
      uint4 v;
      if (F_PP)
        for (u32 si = pf; si != CBH_NONE && !pe; si = uchain_next(t, uload(&t.scope_parent[si]), FLAG_PRIN))
          pe = udir_find(t, CBH_B_PPEXISTS, g_pv, si, 0, v);                            // index.go:999-1021
      for (u32 si = rf; si != CBH_NONE && !re; si = uchain_next(t, uload(&t.scope_parent[si]), FLAG_RES)) {
        if (udir_find(t, CBH_B_RESEXISTS, g_rv, g_k, si, v)) { re = true; break; }      // index.go:966-997
        if (has_rolepol && udir_find(t, CBH_B_RPRES, g_rv, si, 0, v))
          for (u32 k = 0; k < v.y && !re; ++k) re = pmatch(uload(&t.pool[v.x + k]), g_k, g_kbits);
      }
      if (inq) { p_first = pf; r_first = rf; p_exists = pe; r_exists = re; }
    }
  }
  const bool decided = (p_first == CBH_NONE && r_first == CBH_NONE) || (!p_exists && !r_exists);

  // ---- per-action state, bit k = k-th action of the request
#ifdef CBH_PROFILE_CYCLES
  const u64 cyc_pre = __builtin_readcyclecounter();   // only consumed under CBH_F_DEBUG_CYCLES
  const bool dbg = (flags & CBH_F_DEBUG_CYCLES) != 0;
  u64 dbg_t0 = 0, dbg_eval = 0, dbg_n = 0, dbg_t1 = 0, dbg_a = 0, dbg_b = 0, dbg_c = 0, dbg_d = 0;
#define DBG2_T0() do { if (dbg) dbg_t1 = __builtin_readcyclecounter(); } while (0)
#define DBG2_ACC(acc) do { if (dbg) { acc += __builtin_readcyclecounter() - dbg_t1; } } while (0)
#define DBG_T0() do { if (dbg) dbg_t0 = __builtin_readcyclecounter(); } while (0)
#define DBG_ACC(acc) do { if (dbg) { acc += __builtin_readcyclecounter() - dbg_t0; ++dbg_n; } } while (0)
#else
#define DBG_T0() do {} while (0)
#define DBG_ACC(acc) do {} while (0)
#define DBG2_T0() do {} while (0)
#define DBG2_ACC(acc) do {} while (0)
#endif
  AM todo = (valid && !decided) ? all : (AM)0;   // actions still being resolved
  AM eff_allow = 0, eff_deny = 0;             // neither bit set = EFFECT_NO_MATCH so far
  AM st_err = 0, st_unsup = 0;
  // "NO_MATCH" when there is nothing to evaluate (check.go:119-121, 168-170), else the zero EffectInfo (:191)
  write_ps(all, (u32)(decided ? CBH_P_NO_MATCH : CBH_P_EMPTY) << 28, CBH_NONE);

#ifdef CBH_ABLATION   // measurement build (tools/gpu_cycles.py): switch parts of the walk off by flag
  const u32 n_pass = (flags & 0x200u) ? 0u : 2u;   // ablation: skip both passes
  const bool abl_noeval = (flags & 0x400u) != 0, abl_norows = (flags & 0x800u) != 0;
#else
  const u32 n_pass = 2;
#endif
  // The two passes (principal policies, then resource policies: check.go:195) are two instantiations
  // of one body: which pass it is is a compile-time fact, so the principal pass carries no derived-role
  // / role-policy / role-class code and the resource pass none of the principal-only branches.
  auto policy_pass = [&](auto is_res_t) {
    constexpr bool is_res = decltype(is_res_t)::value;
    const u32 first = is_res ? r_first : p_first;
    const u32 flagbit = is_res ? FLAG_RES : FLAG_PRIN;
    const bool exists = is_res ? r_exists : p_exists;
    const u32 gx = is_res ? kind : pid;
    const u32 n_iter = is_res ? role_cnt : (role_cnt ? 1u : 0u);      // check.go:208-213
    // An empty principal chain leaves nothing behind that the resource pass does not overwrite.
    const AM Pm = (is_res || first != CBH_NONE) && n_iter > 0 ? todo : (AM)0;   // actions taking part in this pass
    // roleEffectInfo.Policy: the main policy key if a policy exists at all, else "NO_MATCH" (check.go:216-225)
    const u32 pol_default = exists ? (((u32)(is_res ? CBH_P_RESOURCE : CBH_P_PRINCIPAL) << 28) | first)
                                   : ((u32)CBH_P_NO_MATCH << 28);
    write_ps(Pm, pol_default, CBH_NONE);   // what the first role seeds when nothing matches (check.go:429-431)
    AM rdone = 0;                           // actions that reached ALLOW: they leave the role loop (check.go:433-436)
    bool pend = Pm != 0;
    AM memo_done = 0, memo_val = 0, memo_err = 0;   // per-lane condition outcomes of this pass, bit = record position
    if (want_edr) { DRM(0) = 0; DRM(1) = 0; DRM(2) = 0; }   // likewise for derived-role definitions (done / true / error)

    for (;;) {   // ---- waterfall over groups that share (chain start, version, kind | principal)
      const u64 rem = wave_ballot(pend);
      if (rem == 0) break;
      const u32 lead = first_lane(rem);
      const u32 g_first = wave_readlane(first, lead), g_ver = wave_readlane(r_ver, lead), g_x = wave_readlane(gx, lead);
      const bool ing = pend && first == g_first && r_ver == g_ver && gx == g_x;
      pend = pend && !ing;

      // The directory answers for this group's scopes, kept across the role iterations: every role walks the same
      // chain, so the probe of chain position d is made once (by the first role that gets there) and read back from
      // LDS afterwards - a broadcast read instead of a hash probe with its dependent scalar loads.
      u32 bmemo_n = 0;   // chain positions 0 .. bmemo_n - 1 are in bucket_memo
      for (u32 ri = 0;; ++ri) {   // ---- roles (check.go:208)
        const AM Am = (ing && ri < n_iter) ? (AM)(Pm & todo & ~rdone) : (AM)0;
        if (wave_ballot(Am != 0) == 0) break;
        DBG2_T0();
        RoleSet rs; rs.role = 0; rs.par_off = 0; rs.par_cnt = 0; rs.gbits = 0; rs.classes = 0;
        u32 cls = 63u;
        if (Am != 0) {
          rs.role = ri == 0 ? role0 : ri == 1 ? role1 : b.roles[role_off + ri];
          rs.gbits = F_GLOB ? gbits_of(t, b, DIM_ROLE, rs.role) : 0ull;
          if (is_res) { cls = rs.role < t.K ? (u32)t.role_class[rs.role] : 63u; rs.classes = 1ull << cls; }
          uint4 pv;
          if (has_parents && pr_scope_key != CBH_NONE && dir_find(t, CBH_B_PARENTS, pr_scope_key, rs.role, 0, pv)) {
            rs.par_off = pv.x; rs.par_cnt = pv.y;
            for (u32 k = 0; k < rs.par_cnt; ++k) {   // ancestors are table strings
              const u32 anc = t.pool[rs.par_off + k];
              if (F_GLOB) rs.gbits |= t.gbits[(size_t)DIM_ROLE * t.K + anc];
              rs.classes |= 1ull << (u32)t.role_class[anc];
            }
          }
        }
        // Role classes this wave is walking now (cbh_blob.h CBH_SEC_ROLE_CLASS): a rule record whose
        // class mask misses all of them cannot match any lane and is skipped on the scalar unit.
        // Parent roles widen a lane's role set beyond its own class: no skipping then.
        u64 wave_classes = ~0ull;
        if (is_res && !has_parents) {
          wave_classes = 0;
          bool pendc = Am != 0;
          for (;;) {   // OR over the (few) distinct classes in the wave
            const u64 remc = wave_ballot(pendc);
            if (remc == 0) break;
            const u32 g_cls = wave_readlane(cls, first_lane(remc));
            wave_classes |= 1ull << g_cls;
            pendc = pendc && cls != g_cls;
          }
        }
        AM has_allow = 0;
        AM S = Am;   // actions still walking the scope chain for this role
        u32 site_ctr = 0, dr_site_ctr = 0;
        DBG2_ACC(dbg_a);

        // effect events of this role iteration (fold of check.go:382-442, applied as they happen)
        auto role_deny = [&](AM mask, u32 polw, u32 si) {
          const AM seed = mask & ~(eff_allow | eff_deny);   // still NO_MATCH: this role's DENY seeds the result
          eff_deny |= seed;
          write_ps(seed, polw, si);
          S &= ~mask;
        };
        auto role_allow = [&](AM mask, u32 si) {           // first independent ALLOW wins (check.go:433-436)
          eff_allow |= mask; eff_deny &= ~mask; rdone |= mask;
          write_ps(mask, pol_default, si);
          S &= ~mask;
        };
        auto strict_deny = [&](AM mask, u32 polw, u32 si) { // evaluation error in strict mode (check.go:353-356, 371-374)
          eff_deny |= mask; eff_allow &= ~mask; todo &= ~mask;
          write_ps(mask, polw, si);
          S &= ~mask;
        };
        auto take_status = [&](AM mask) {                    // attribute VM status bits to the actions served
          if (L.status & CBH_ST_CEL_ERROR) st_err |= mask;
          if (L.status & CBH_ST_UNSUPPORTED) st_unsup |= mask;
          L.status = 0;
        };
        // ---- trace pass: the programs that keep expression identities (cbh_blob.h CBH_SEC_TRACE_*)
        auto trace_run = [&](u32 tpc, bool on, u32 tctx, u64 tmask) -> int {   // tmask: the actions an output belongs to
          const u32 r = run_uniform_trace(c.ka_mem, lds_of(c), L.req, L.edr, L.edr_err, tpc, on, tctx, tmask, L.edr_errmask);
          if (on) { L.status |= r >> 8; return (int)(r & 0xFF); }
          return 0;
        };
        // every variable of the params set is evaluated, referenced or not (check.go:651-677); what they yield is not
        // needed here - the condition programs carry the definitions inline - only the errors they raise
        auto trace_vars = [&](u32 off, u32 cnt, bool on, u32 tctx) {
          for (u32 i = 0; i < cnt; ++i) (void)trace_run(uload(&t.trace_pool[off + i]), on, tctx, 0);
        };
        auto trace_ctx = [&](u32 site) -> u32 {
          if (ri > 0xFFu || site > 0xFFFu) L.status |= CBH_ST_UNSUPPORTED;   // beyond the record's fields
          return ((is_res ? 1u : 0u) << 4) | ((ri & 0xFFu) << 12) | ((site & 0xFFFu) << 20);
        };

        u32 chain_pos = 0;
        for (u32 si = g_first; si != CBH_NONE; si = uchain_next(t, uload(&t.scope_parent[si]), flagbit), ++chain_pos) {   // check.go:231
          if (wave_ballot(S != 0) == 0) break;
          DBG2_T0();
          uint4 bucket; bucket.x = bucket.y = bucket.z = bucket.w = 0;
          bool have_bucket;
          if (chain_pos < bmemo_n) {
            CBH_L const u32* e = bucket_memo + chain_pos * 5u;
            bucket.x = uniform(e[0]); bucket.y = uniform(e[1]); bucket.z = uniform(e[2]); bucket.w = uniform(e[3]);
            have_bucket = uniform(e[4]) != 0;
          } else {
            have_bucket = is_res ? udir_find(t, CBH_B_RESOURCE, g_ver, g_x, si, bucket)
                                 : udir_find(t, CBH_B_PRINCIPAL, g_ver, si, g_x, bucket);   // resource version: check.go:294
            if (chain_pos == bmemo_n && chain_pos < CBH_BUCKET_MEMO) {
              CBH_L u32* e = bucket_memo + chain_pos * 5u;
              if (c.tid == 0) { e[0] = bucket.x; e[1] = bucket.y; e[2] = bucket.z; e[3] = bucket.w; e[4] = have_bucket ? 1u : 0u; }
              ++bmemo_n;
            }
          }

          DBG2_ACC(dbg_b);
          if (is_res && want_edr) {   // derived roles of this scope's resource policy (check.go:237-282)
            // Whether a definition applies (parent roles) and what its condition yields depend on the
            // request only, not on the role being walked: the outcome of the first DRM_SITES
            // definitions met in this group's walk is kept per lane (LDS) and replayed for the
            // request's other roles - the reference computes them once per scope too (check.go:237).
            u64 m = 0, derr_names = 0; bool derr = false;
            if (have_bucket) {
              for (u32 d = bucket.z; d < bucket.z + bucket.w; ++d) {
                const TblDr dr = uload_rec<TblDr>(t.dr, d);
                const u32 dsite = dr_site_ctr++;
                const AM dbit = dsite < AM_BITS ? (AM)((AM)1 << dsite) : (AM)0;
                const bool act = S != 0;
                const bool hit = act && (DRM(0) & dbit) != 0;
                const bool fresh = act && !hit;
                bool applies = false;
                if (fresh) applies = dr.parents_cnt == CBH_NONE ||
                    lane_has_parent_role(t, b, dr.parents_off, dr.parents_cnt, role_off, role_cnt, pr_scope_key, has_parents);
                int r = 1;
                if (TRACE) {
                  if (wave_ballot(applies) != 0) {   // check.go:251-270: the definition's variables, then its condition
                    const TblTraceCond td = uload_rec<TblTraceCond>(t.trace_dr, d);
                    const u32 tctx = trace_ctx(0);
                    trace_vars(td.vars_off, td.vars_cnt, applies, tctx);
                    if (td.cond != CBH_NONE) r = trace_run(td.cond, applies, tctx, 0);
                  }
                } else
                if (dr.cond != CBH_NONE && wave_ballot(applies) != 0) r = eval_cond<GENERIC>(c, L, dr.cond, applies);
                if (fresh) {
                  if (!(L.status & CBH_ST_UNSUPPORTED)) {
                    DRM(0) |= dbit;
                    if (applies && r == 1) DRM(1) |= dbit;
                    if (applies && (L.status & CBH_ST_CEL_ERROR)) DRM(2) |= dbit;
                  }
                  if (applies) { if (r == 2) { derr = true; derr_names |= 1ull << dr.name; } else if (r == 1) m |= 1ull << dr.name; take_status(S); }
                }
                if (hit) {
                  if (DRM(2) & dbit) { L.status |= CBH_ST_CEL_ERROR; if (strict) { derr = true; derr_names |= 1ull << dr.name; } take_status(S); }   // an error is a failed definition in strict mode
                  else if (DRM(1) & dbit) m |= 1ull << dr.name;
                }
              }
            }
            if (S != 0) { L.edr = m; L.edr_err = derr; edr_acc |= m; if (TRACE) L.edr_errmask = derr_names; }
          }

          // role policies exist at (version, scope) at all?  CBH_B_RPRES is keyed by exactly that: without an entry no role
          // has a role-policy bucket here, the whole block below would find nothing - one probe instead of a scan of
          // the bucket's records plus a probe per distinct role
          uint4 rpres_here;
          const bool rolepol_here = is_res && has_rolepol && udir_find(t, CBH_B_RPRES, g_ver, si, 0, rpres_here);
          if (rolepol_here) {
            // baseBM of Index.Query (index.go:250-305): the scope yields synthetic DENYs only if SOME binding at
            // (version, scope) - a rule of the resource policy or a role-policy rule - matches the resource and
            // one of [role] ++ ancestors; an empty base returns before appendRolePolicyDenies.
            bool base = false;
            if (have_bucket)
              for (u32 row = bucket.x; row < bucket.x + bucket.y; ++row) {
                const TblRow rw = uload_rec<TblRow>(t.rows, 2 * row);
                const bool m = rec_role_match(rw, row, rs, S != 0 && !base);
                base = base || m;
              }
            for (u32 k = 0;; ++k) {
              const bool P = S != 0 && !base && k <= rs.par_cnt;
              if (wave_ballot(P) == 0) break;
              const u32 srole = P ? (k == 0 ? rs.role : t.pool[rs.par_off + k - 1]) : 0;
              bool pend2 = P;
              for (;;) {
                const u64 rem2 = wave_ballot(pend2);
                if (rem2 == 0) break;
                const u32 g_sr = wave_readlane(srole, first_lane(rem2));
                const bool in2 = pend2 && srole == g_sr;
                pend2 = pend2 && !in2;
                uint4 rp;
                if (!udir_find(t, CBH_B_ROLEPOL, g_ver, si, g_sr, rp)) continue;
                for (u32 row = rp.x; row < rp.x + rp.y; ++row) {
                  const TblRp rr = uload_rec<TblRp>(t.rprows, row);
                  if (in2 && pmatch(rr.resource, kind, KIND_BITS())) base = true;
                }
              }
            }
            // synthetic DENYs from the role policies of [role] ++ ancestors (index.go:352-530)
            for (u32 k = 0;; ++k) {
              const bool P = S != 0 && base && k <= rs.par_cnt;
              if (wave_ballot(P) == 0) break;
              const u32 srole = P ? (k == 0 ? rs.role : t.pool[rs.par_off + k - 1]) : 0;
              bool pend2 = P;
              for (;;) {   // waterfall over the distinct roles in flight
                const u64 rem2 = wave_ballot(pend2);
                if (rem2 == 0) break;
                const u32 g_sr = wave_readlane(srole, first_lane(rem2));
                const bool in2 = pend2 && srole == g_sr;
                pend2 = pend2 && !in2;
                uint4 rp;
                if (!udir_find(t, CBH_B_ROLEPOL, g_ver, si, g_sr, rp)) continue;
                const u32 rp_pol = ((u32)CBH_P_TABLE << 28) | rp.z;
                AM any_mask = 0;   // actions allowed (subject to conditions) by some rule for this resource
                AM out_only = 0;   // ... by an output-only rule that shares its cache key with a conditional one (cbh_blob.h CBH_RP_F_*)
                for (u32 row = rp.x; row < rp.x + rp.y; ++row) {
                  const TblRp rr = uload_rec<TblRp>(t.rprows, row);
                  if (!in2 || !pmatch(rr.resource, kind, KIND_BITS())) continue;
                  AM ma = 0;
                  for (u32 a = 0; a < (rr.allow_cnt & CBH_RP_CNT_MASK); ++a) ma |= match_actions(uload(&t.pool[rr.allow_off + a]));
                  any_mask |= ma;
                  if ((rr.allow_cnt & (CBH_RP_F_OUTPUT_ONLY | CBH_RP_F_SHARES_KEY)) == (CBH_RP_F_OUTPUT_ONLY | CBH_RP_F_SHARES_KEY)) out_only |= ma;
                }
                // no binding for the resource, or no allow-action matched (index.go:436-461)
                AM deny = in2 ? (AM)(S & ~any_mask) : (AM)0;
                if (want_ep && deny != 0) ep_note(rp.z & 0x0FFFFFFFu);   // the synthetic DENY is a binding of the role policy (check.go:302-304)
                const u32 site_base = site_ctr;   // trace pass: a visit's place in the walk = its row's place in the bucket
                AM cond_seen = 0; int cond_r = 0;   // trace pass: actions a key-sharing conditional rule was evaluated for, and what it gave
                // (trace pass: the output-only rules that share a key come last, when every conditional rule has been seen;
                //  they have no effect, and a visit's place in the log is its row's place, so nothing else moves)
                for (u32 late = 0; late < (TRACE ? 2u : 1u); ++late)
                for (u32 row = rp.x; row < rp.x + rp.y; ++row) {
                  const TblRp rr = uload_rec<TblRp>(t.rprows, row);
                  if (TRACE && (((rr.allow_cnt & (CBH_RP_F_OUTPUT_ONLY | CBH_RP_F_SHARES_KEY)) == (CBH_RP_F_OUTPUT_ONLY | CBH_RP_F_SHARES_KEY)) ? 1u : 0u) != late) continue;
                  // the reference visits a matched role-policy rule only if it has a condition (as the synthetic DENY row
                  // none(cond)) or an output expression (index.go:436-530)
                  TblTraceRp tp{};
                  if (TRACE) tp = uload_rec<TblTraceRp>(t.trace_rp, row);
                  const bool has_out = TRACE && (tp.out_activated != CBH_NONE || tp.out_not_met != CBH_NONE);
                  if (rr.cond == CBH_NONE && !has_out) continue;
                  AM mm = 0;
                  if (in2 && pmatch(rr.resource, kind, KIND_BITS())) {
                    for (u32 a = 0; a < (rr.allow_cnt & CBH_RP_CNT_MASK); ++a) mm |= match_actions(uload(&t.pool[rr.allow_off + a]));
                    mm &= S & ~deny;
                  }
                  if (wave_ballot(mm != 0) == 0) continue;
                  if (want_ep && mm != 0) ep_note(rp.z & 0x0FFFFFFFu);
                  const bool shares = (rr.allow_cnt & CBH_RP_F_SHARES_KEY) != 0;
                  // an action at or behind the first one an output-only rule of the same key is visited for finds
                  // "satisfied" cached (check.go:324): the synthetic DENY would fire whatever the condition says
                  if (shares && rr.cond != CBH_NONE && out_only != 0) st_unsup |= (AM)(mm & ~(AM)((out_only & (AM)(0 - out_only)) - 1));
                  int r = 1;
                  if (TRACE) {
                    const u32 tctx = trace_ctx(site_base + (row - rp.x));
                    trace_vars(tp.vars_off, tp.vars_cnt, mm != 0, tctx);
                    if (rr.cond != CBH_NONE) {
                      r = trace_run(tp.cond, mm != 0, tctx, 0);
                      if (shares && mm != 0) { if (cond_seen == 0) cond_r = r; cond_seen |= mm; }
                    }
                    // The rule's outputs as its author wrote them (the synthetic row swaps them together with the condition).
                    // An output-only rule reached after a conditional one of its key sees that rule's cached outcome
                    // none(condition): per action, since which of the two came first is a matter of the action order.
                    AM act = (mm != 0 && r == 1) ? mm : (AM)0, notmet = (mm != 0 && r == 0) ? mm : (AM)0;
                    bool lost = false;
                    if (rr.cond == CBH_NONE && shares && cond_seen != 0) {
                      const AM first = (AM)(cond_seen & (AM)(0 - cond_seen));
                      const AM behind = (AM)(mm & ~(AM)((first << 1) - 1));       // actions after the conditional rule's first visit
                      lost = (mm & first) != 0 || (behind != 0 && cond_r == 2);    // the same action: rule order and role order decide
                      if (cond_r == 1) { act = (AM)(mm & ~behind & ~first); notmet = behind; }   // condition held: none(..) is false
                      else act = (AM)(mm & ~first);
                    }
                    const u32 st0 = L.status;
                    if (tp.out_activated != CBH_NONE && wave_ballot(act != 0) != 0) (void)trace_run(tp.out_activated, act != 0, tctx, (u64)act);
                    if (tp.out_not_met != CBH_NONE && wave_ballot(notmet != 0) != 0) (void)trace_run(tp.out_not_met, notmet != 0, tctx, (u64)notmet);
                    if (lost || (L.status & ~st0 & CBH_ST_UNSUPPORTED) != 0) trace_log(o, L.req, CBH_TR_INCOMPLETE | tctx, 0, 1, 0, (u64)mm);
                    L.status = st0;
                  } else
                  r = eval_cond<GENERIC>(c, L, rr.cond, mm != 0);   // synthetic row = DENY if none(cond)
                  if (mm != 0) {
                    take_status(mm);
                    if (rr.cond != CBH_NONE) {
                      if (r == 2) strict_deny(mm, rp_pol, si);
                      else if (r == 0) deny |= mm;
                    }
                  }
                }
                if (TRACE) site_ctr += rp.y;
                deny &= S;
                if (deny != 0) role_deny(deny, rp_pol, si);   // check.go:395-403
              }
            }
          }

          if (have_bucket) {   // ---- regular rows of the bucket, in binding order (check.go:295-414)
#ifdef CBH_ABLATION
            if (abl_norows) bucket.y = 0;
#endif
            for (u32 row = bucket.x; row < bucket.x + bucket.y; ++row) {
              DBG2_T0();
              const TblRow rw = uload_rec<TblRow>(t.rows, 2 * row);
              const LeafRec lf = uload_rec<LeafRec>(t.rows, 2 * row + 1);   // issued with the first half: no dependent load for a leaf condition
              const u32 site = site_ctr++;   // position of the record in this group's walk: the same for every role
              if ((((u64)rw.rm_lo | ((u64)rw.rm_hi << 32)) & wave_classes) == 0) continue;   // no lane's role can match
              const u32 e = rw.flags & 3u;
              // a record = roles x actions of one rule (cbh_blob.h): the lists are wave-uniform
              AM mrow = 0;
              constexpr u32 BY_CLASS = CBH_ROW_F_ROLE_BY_CLASS | CBH_ROW_F_ACTION_BY_CLASS;
              if (is_res && F_MAX4 && (rw.flags & BY_CLASS) == BY_CLASS) {
                // both lists are literals with a class: two mask tests, no pattern reference is read
                const u64 am = (u64)rw.am_lo | ((u64)rw.am_hi << 32);
                const bool rmatch = S != 0 && ((((u64)rw.rm_lo | ((u64)rw.rm_hi << 32)) & rs.classes) != 0);
                const AM mact = (AM)(((am >> ac0) & 1ull) | (((am >> ac1) & 1ull) << 1) | (((am >> ac2) & 1ull) << 2) | (((am >> ac3) & 1ull) << 3));
                mrow = rmatch ? (AM)(mact & S) : (AM)0;
              } else {
                const TblRowPat pt = uload_rec<TblRowPat>(t.rowpat, row);
                const u32 n_act = pt.counts & 0xFFFFu;   // 0 = a single inline reference
                const bool rmatch = is_res ? pat_role_match(rw, pt, rs, S != 0) : (S != 0 && pmatch(pt.resource, kind, KIND_BITS()));
                if (rmatch) {
                  if (rw.flags & CBH_ROW_F_ACTION_LIST) {
                    for (u32 i = 0; i < n_act; ++i) mrow |= match_actions(uload(&t.pool[pt.action + i]));
                  } else {
                    mrow = match_actions(pt.action);
                    if (n_act > 1) mrow |= match_actions(pt.a1);
                    if (n_act > 2) mrow |= match_actions(pt.a2);
                  }
                  mrow &= S;
                }
              }
              // every matched row is evaluated, also an ALLOW after an ALLOW that already fired: the
              // reference does the same (check.go:295-414), and its errors belong in evaluation_errors
              const AM need = mrow;
              DBG2_ACC(dbg_c);
              if (wave_ballot(need != 0) == 0) continue;
              const bool m = need != 0;
              if (want_ep && m) ep_note(rw.policy);   // the binding is iterated: its policy set is in effect (check.go:302-304)
              // A request meets the same record once per role it holds; its conditions read only the
              // request (and this scope's derived roles), so the first outcome is kept per lane for the
              // first 64 records of the walk and replayed - including the error status - afterwards.
              const AM sbit = site < AM_BITS ? (AM)((AM)1 << site) : (AM)0;
              const bool hit = !TRACE && m && (memo_done & sbit) != 0;   // the trace pass logs every visit
              const bool mev = m && !hit;
              int r = 1;
              if (TRACE) {
                const TblTraceRow tr = uload_rec<TblTraceRow>(t.trace_rows, row);
                const u32 tctx = trace_ctx(site);
                trace_vars(tr.vars_off, tr.vars_cnt, m, tctx);                             // check.go:306-321
                bool drfail = false;
                if (rw.drcond != CBH_NONE) {                                               // check.go:328-366
                  trace_vars(tr.drvars_off, tr.drvars_cnt, m, tctx);
                  r = trace_run(tr.drcond, m, tctx, 0);
                  drfail = m && r == 0;
                }
                const bool m2 = m && r == 1;
                if (rw.cond != CBH_NONE && wave_ballot(m2) != 0) {                         // check.go:368-380
                  const int r2 = trace_run(tr.cond, m2, tctx, 0);
                  if (m2) r = r2;
                }
                // outputs (check.go:383-411): one record per visit, the actions it stands for in the mask.  (A rule with
                // outputs keeps one record per rule-table row - blob.py add_bucket_rows - so a visit here is a visit there.)
                // An output expression outside the device subset costs the request its outputs, not its decision or errors.
                const bool act = m && r == 1, notmet = m && r == 0;
                const u32 st0 = L.status;
                if (tr.out_activated != CBH_NONE && wave_ballot(act) != 0) (void)trace_run(tr.out_activated, act, tctx, (u64)mrow);
                if (tr.out_not_met != CBH_NONE && wave_ballot(notmet) != 0)
                  (void)trace_run(tr.out_not_met, notmet, tctx | (drfail ? CBH_TR_DRFAIL : 0u), (u64)mrow);
                if ((L.status & ~st0 & CBH_ST_UNSUPPORTED) != 0) trace_log(o, L.req, CBH_TR_INCOMPLETE | tctx, 0, 1, 0, (u64)mrow);
                L.status = st0;
              } else
#ifdef CBH_ABLATION
              if (abl_noeval) {} else
#endif
              if ((rw.cond != CBH_NONE || rw.drcond != CBH_NONE) && wave_ballot(mev) != 0) {
                DBG_T0();
                if (rw.drcond != CBH_NONE) r = eval_cond<GENERIC>(c, L, rw.drcond, mev);   // check.go:328-366
                const bool m2 = mev && r == 1;
                if (rw.cond != CBH_NONE && wave_ballot(m2) != 0) {                         // check.go:368-380
                  const int r2 = (rw.flags & CBH_ROW_F_LEAF_EMBEDDED) ? eval_cond_rec<GENERIC>(c, L, rw.cond, lf, m2)
                                                                      : eval_cond<GENERIC>(c, L, rw.cond, m2);
                  if (m2) r = r2;
                }
                DBG_ACC(dbg_eval);
                if (mev && !(L.status & CBH_ST_UNSUPPORTED)) {
                  memo_done |= sbit;
                  if (r == 1) memo_val |= sbit;
                  if (L.status & CBH_ST_CEL_ERROR) memo_err |= sbit;
                }
              }
              DBG2_T0();
              if (hit) {
                r = (memo_val & sbit) ? 1 : 0;
                if (memo_err & sbit) { L.status |= CBH_ST_CEL_ERROR; if (strict) r = 2; }   // an error is a DENY in strict mode
              }
              if (m) {
                take_status(need);
                if (r == 2) strict_deny(need, ((u32)CBH_P_TABLE << 28) | rw.policy, si);
                else if (r == 1) {
                  if (e == CBH_EFFECT_ALLOW) has_allow |= mrow;
                  else if (e == CBH_EFFECT_DENY) role_deny(mrow, pol_default, si);
                }
              }
              DBG2_ACC(dbg_d);
            }
          }

          const AM ha = has_allow & S;   // check.go:416-425
          if (ha != 0) {
            const u32 sp = (uload(&t.scope_flags[si]) >> 2) & 3u;
            if (sp == SP_REQUIRE_CONSENT) has_allow &= ~ha;
            else if (sp == SP_OVERRIDE_PARENT) role_allow(ha, si);
          }
        }
      }
    }
    // a definitive principal-policy result ends the action (check.go:445-448)
    todo &= ~(eff_allow | eff_deny);
  };
  if (n_pass) {
    if constexpr (F_PP) policy_pass(CbhPassPrincipal{});
    policy_pass(CbhPassResource{});
  }

#ifdef CBH_PROFILE_CYCLES
  if ((flags & CBH_F_DEBUG_CYCLES) && want_ps && act_cnt >= 3) {
    // profiling aid: policy words of the first three actions <- cycles spent in the preamble,
    // in the two policy passes, and the wave's start time (low 32 bits)
    const u64 cyc_end = __builtin_readcyclecounter();
    PS_POL(0) = (u32)(cyc_pre - cyc_start); PS_POL(1) = (u32)(cyc_end - cyc_pre); PS_POL(2) = (u32)dbg_eval; PS_POL(3) = (u32)dbg_n;
    PS_SCP(0) = (u32)dbg_a; PS_SCP(1) = (u32)dbg_b; PS_SCP(2) = (u32)dbg_c; PS_SCP(3) = (u32)dbg_d;
  }
#endif
  // A request with exactly four actions on a 4-tuple boundary (the usual batch shape) writes each
  // output array with ONE full-width store per lane: whole cache lines per wave instead of byte
  // stores scattered four bytes apart.
  // a table with variables or outputs: this walk cannot tell which inputs the trace pass has something for (cerbos_hip.h)
  const u32 st_ok = (!TRACE && (t.flags & CBH_MF_TRACE_ALL)) ? CBH_ST_WANTS_TRACE : CBH_ST_OK;
  const bool packed = valid && act_cnt == 4 && (act_off & 3u) == 0;
  if (packed) {
    struct __attribute__((aligned(16))) u32x4 { u32 x, y, z, w; };
    u32 e4 = 0, s4 = 0;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
      const AM bit = (AM)1 << k;
      e4 |= (u32)((eff_allow & bit) ? CBH_EFFECT_ALLOW : CBH_EFFECT_DENY) << (8 * k);
      s4 |= (u32)((st_unsup & bit) ? CBH_ST_UNSUPPORTED : ((st_err & bit) ? CBH_ST_CEL_ERROR : st_ok)) << (8 * k);
    }
    if (o.edr) o.edr[req] = edr_acc;
    *(CBH_G u32*)(o.effect + act_off) = e4;
    if (o.status) *(CBH_G u32*)(o.status + act_off) = s4;
    if (o.policy) { u32x4 v; v.x = PS_POL(0); v.y = PS_POL(1); v.z = PS_POL(2); v.w = PS_POL(3); *(CBH_G u32x4*)(o.policy + act_off) = v; }
    if (o.scope) { u32x4 v; v.x = PS_SCP(0); v.y = PS_SCP(1); v.z = PS_SCP(2); v.w = PS_SCP(3); *(CBH_G u32x4*)(o.scope + act_off) = v; }
  } else if (valid) {
    if (o.edr) o.edr[req] = edr_acc;
    if (want_ps) {
#pragma unroll
      for (u32 k = 0; k < 4; ++k) {
        if (k < act_cnt) {
          if (o.policy) o.policy[act_off + k] = PS_POL(k);
          if (o.scope) o.scope[act_off + k] = PS_SCP(k);
        }
      }
    }
    for (u32 k = 0; k < act_cnt; ++k) {
      const AM bit = (AM)1 << k;
      o.effect[act_off + k] = (u8)((eff_allow & bit) ? CBH_EFFECT_ALLOW : CBH_EFFECT_DENY);   // NO_MATCH -> DENY (check.go:451-453)
      if (o.status) o.status[act_off + k] = (u8)((st_unsup & bit) ? CBH_ST_UNSUPPORTED : ((st_err & bit) ? CBH_ST_CEL_ERROR : st_ok));
    }
  }
#undef DRM
#undef PS_POL
#undef KIND_BITS
#undef PS_SCP
}

// Dynamic LDS of the kernels = the column cache: CBH_CC_DWORDS(n_cached, packed) dwords.
#ifndef CBH_HOSTSIM
extern __shared__ __attribute__((aligned(16))) unsigned char cbh_dyn_lds[];
#else
#ifndef CBH_HOSTSIM_LDS_WAVES
#define CBH_HOSTSIM_LDS_WAVES 1   /* (the engine simulation's four-wave build: 4) */
#endif
static unsigned char cbh_dyn_lds[(CBH_HOSTSIM_LDS_WAVES) * (CBH_CACHE_COLS * CBH_BLOCK * 12 + CBH_ARENA_ENTRIES * CBH_BLOCK * 9 + 16 * CBH_BLOCK * 4 + 8 * CBH_BLOCK * 4 + 16 * CBH_BLOCK * 8 + 16 * CBH_BLOCK * 8 + 3 * 4096 + 4 * 1024 + 32 + 2560 + 16 * CBH_BLOCK * 8)];   // + the flat / walk2 kernels' chain scratch, per-action notes, site results, class tables, the trail's touches
#endif
__device__ __forceinline__ u32 cached_columns(const KernelArgs* ka) {
  const u32 n = ka->b.n_columns;
  return n < CBH_CACHE_COLS ? n : CBH_CACHE_COLS;
}

// GENERIC instantiation: operand stack, locals and iteration slots in LDS, laid out [slot][lane].
template <typename AM, int FEAT>
__device__ __forceinline__ void generic_kernel_body(const KernelArgs& a, const KernelArgs* ka) {
  __shared__ u64 s_val[CBH_STACK_DEPTH * CBH_BLOCK];
  __shared__ u64 l_val[CBH_MAX_LOCALS * CBH_BLOCK];
  __shared__ u64 it_cont[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u32 it_idx[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u32 it_state[CBH_MAX_ITERS * ((FEAT & CBH_FEAT_TRACE) ? 3 : 1) * CBH_BLOCK];   // trace pass: + the absorbed error per slot (cbh_interp.h IT_ERR_*)
  __shared__ u8 s_tag[CBH_STACK_DEPTH * CBH_BLOCK];
  __shared__ u8 l_tag[CBH_MAX_LOCALS * CBH_BLOCK];
  // The interpreter runs every lane of the wave through a program, lanes that are not evaluating it
  // included, and those read whatever their operand slots hold.  LDS is not cleared between
  // workgroups: start every slot of this lane as an error value / empty container, so that a stale
  // "string id" or "list offset" left by another kernel can never be dereferenced.
  {
    const u32 tid = threadIdx.x;
    for (u32 k = 0; k < CBH_STACK_DEPTH; ++k) { s_tag[k * CBH_BLOCK + tid] = CBH_T_ERR; s_val[k * CBH_BLOCK + tid] = 0; }
    for (u32 k = 0; k < CBH_MAX_LOCALS; ++k) { l_tag[k * CBH_BLOCK + tid] = CBH_T_ERR; l_val[k * CBH_BLOCK + tid] = 0; }
    for (u32 k = 0; k < CBH_MAX_ITERS; ++k) { it_cont[k * CBH_BLOCK + tid] = 0; it_idx[k * CBH_BLOCK + tid] = 0; it_state[k * CBH_BLOCK + tid] = 0; }
  }
  const u32 ncc = cached_columns(&a);
  Ctx c{a.t, a.b, a.now_ns, a.flags, threadIdx.x,
        (CBH_L u64*)s_val, (CBH_L u8*)s_tag, (CBH_L u64*)l_val, (CBH_L u8*)l_tag,
        (CBH_L u64*)it_cont, (CBH_L u32*)it_idx, (CBH_L u32*)it_state,
        (CBH_L u32*)cbh_dyn_lds, ncc, ka};
  check_body<true, AM, FEAT>(a, c);
}

// Leaf-only instantiation: no operand stack, no interpreter call.
template <typename AM, int FEAT>
__device__ __forceinline__ void leaf_kernel_body(const KernelArgs& a, const KernelArgs* ka) {
  const u32 ncc = cached_columns(&a);
  Ctx c{a.t, a.b, a.now_ns, a.flags, threadIdx.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
        (CBH_L u32*)cbh_dyn_lds, ncc, ka};
  check_body<false, AM, FEAT>(a, c);
}

// A batch of 1M tuples is ~3.9k waves for 1024 SIMDs: holding the leaf kernels to 128 VGPRs lets all
// of them be resident at once (4 waves per SIMD) instead of running in two rounds.
#ifndef CBH_HOSTSIM
#define CBH_FOUR_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))
#ifdef CBH_GENERIC_WPE   /* lab: occupancy target of the kernels that carry the operand-stack interpreter */
#define CBH_GENERIC_WAVES __attribute__((amdgpu_waves_per_eu(CBH_GENERIC_WPE, CBH_GENERIC_WPE)))
#else
#define CBH_GENERIC_WAVES
#endif
#else
#define CBH_FOUR_WAVES
#define CBH_GENERIC_WAVES
#endif
// The host picks by table - every program a fused leaf / leaf tree?  which features does it use
// (CBH_FEAT_*: a table without derived roles / role policies / parent roles gets a kernel that does
// not carry their code or registers) - and by batch (no request with more than 32 actions -> 32-bit
// action masks).  Name = cbh_check_kernel[_leaf][_a32[_f<feature bits>]]; no feature suffix = everything.
#define CBH_DEFINE_CHECK_KERNELS(AMT, FEAT, SUF)                                                                              \
  __global__ __launch_bounds__(CBH_BLOCK) CBH_GENERIC_WAVES void cbh_check_kernel##SUF(const KernelArgs a, const KernelArgs* __restrict__ ka) { \
    generic_kernel_body<AMT, FEAT>(a, ka);                                                                                    \
  }                                                                                                                           \
  __global__ __launch_bounds__(CBH_BLOCK) CBH_FOUR_WAVES void cbh_check_kernel_leaf##SUF(const KernelArgs a, const KernelArgs* __restrict__ ka) { \
    leaf_kernel_body<AMT, FEAT>(a, ka);                                                                                       \
  }
CBH_DEFINE_CHECK_KERNELS(u32, 0, _a32_f0)                                                   // plain policies, literal patterns
CBH_DEFINE_CHECK_KERNELS(u32, CBH_FEAT_DERIVED_ROLES, _a32_f1)                              // + derived roles
CBH_DEFINE_CHECK_KERNELS(u32, CBH_FEAT_GLOBS, _a32_f4)                                      // plain + glob patterns
CBH_DEFINE_CHECK_KERNELS(u32, CBH_FEAT_DERIVED_ROLES | CBH_FEAT_GLOBS, _a32_f5)             // derived roles + globs
CBH_DEFINE_CHECK_KERNELS(u32, CBH_FEAT_ALL, _a32)   // everything: + role policies, parent roles, principal policies
CBH_DEFINE_CHECK_KERNELS(u64, CBH_FEAT_ALL, )       // everything, > 32 actions
// the trace pass (cbh_trace_batch): everything, every condition through its trace program
__global__ __launch_bounds__(CBH_BLOCK) void cbh_trace_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  generic_kernel_body<u64, CBH_FEAT_ALL | CBH_FEAT_TRACE>(a, ka);
}
// cbh_check_batch_trail: everything, and which policies' bindings the walk iterates (ep_mark)
__global__ __launch_bounds__(CBH_BLOCK) void cbh_check_trail_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  generic_kernel_body<u64, CBH_FEAT_ALL | CBH_FEAT_TRAIL>(a, ka);
}
// the leaf kernels of the four common table classes once more for batches with <= 4 actions per request
#define CBH_DEFINE_LEAF_A4(FEAT, SUF)                                                                                          \
  __global__ __launch_bounds__(CBH_BLOCK) CBH_FOUR_WAVES void cbh_check_kernel_leaf##SUF(const KernelArgs a, const KernelArgs* __restrict__ ka) { \
    leaf_kernel_body<u32, (FEAT) | CBH_FEAT_MAX4>(a, ka);                                                                      \
  }
CBH_DEFINE_LEAF_A4(0, _a4_f0)
CBH_DEFINE_LEAF_A4(CBH_FEAT_DERIVED_ROLES, _a4_f1)
CBH_DEFINE_LEAF_A4(CBH_FEAT_GLOBS, _a4_f4)
CBH_DEFINE_LEAF_A4(CBH_FEAT_DERIVED_ROLES | CBH_FEAT_GLOBS, _a4_f5)

typedef void (*cbh_check_kernel_fn)(const KernelArgs, const KernelArgs*);
// the instantiation for a table (its meta flags, number of derived-role records) and a batch
static inline cbh_check_kernel_fn cbh_pick_check_kernel(u32 table_flags, u32 n_derived_roles, bool has_globs, u32 max_actions) {
  const int g = (table_flags & CBH_MF_HAS_GENERIC_PROGRAMS) ? 1 : 0;
  if (max_actions > 32) return g ? cbh_check_kernel : cbh_check_kernel_leaf;
  if (table_flags & (CBH_MF_HAS_ROLE_POLICIES | CBH_MF_HAS_PARENT_ROLES | CBH_MF_HAS_PRINCIPAL_POLICIES))
    return g ? cbh_check_kernel_a32 : cbh_check_kernel_leaf_a32;
  const bool dr = n_derived_roles || (table_flags & CBH_MF_USES_RUNTIME_EDR);
  static const cbh_check_kernel_fn tab[2][2][2] = {   // [globs][derived roles][generic]
      {{cbh_check_kernel_leaf_a32_f0, cbh_check_kernel_a32_f0}, {cbh_check_kernel_leaf_a32_f1, cbh_check_kernel_a32_f1}},
      {{cbh_check_kernel_leaf_a32_f4, cbh_check_kernel_a32_f4}, {cbh_check_kernel_leaf_a32_f5, cbh_check_kernel_a32_f5}}};
  if (!g && max_actions <= 4) {
    static const cbh_check_kernel_fn tab4[2][2] = {{cbh_check_kernel_leaf_a4_f0, cbh_check_kernel_leaf_a4_f1},
                                                   {cbh_check_kernel_leaf_a4_f4, cbh_check_kernel_leaf_a4_f5}};
    return tab4[has_globs ? 1 : 0][dr ? 1 : 0];
  }
  return tab[has_globs ? 1 : 0][dr ? 1 : 0][g];
}
