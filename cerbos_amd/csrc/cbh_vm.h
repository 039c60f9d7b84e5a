// Device-side value model and CEL bytecode interpreter for the decision kernel.
// CDNA4 / gfx950 only.  One lane evaluates one (principal, resource, action) tuple; the
// operand stack, locals and iteration slots live in LDS laid out [slot][lane] so a
// wave's accesses to one slot hit 64 consecutive banks-worth of addresses (conflict free).
//
// Semantics restate the CEL behaviour the reference gets from cel-go v0.30.0 through
// internal/conditions/cel.go:65-107 (see oracle/celeval.py for the CPU restatement and
// SURVEY.md Appendix B for the rules): cross-type numeric comparison, int64 overflow
// errors, error absorption in && / ||, missing-key errors, leaf errors -> false.
#pragma once
#ifndef CBH_HOSTSIM
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "../../include/cerbos_hip.h"
#include "cbh_blob.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

// One wave per workgroup: the kernels need no cross-wave cooperation, and with one lane per
// request a 1M-tuple batch is only ~4k waves, so single-wave groups spread evenly over 256 CUs.
#define CBH_BLOCK 64
#define CBH_STACK_DEPTH 10
#define CBH_MAX_LOCALS 4
#define CBH_MAX_ITERS 2

// Address spaces.  The structs below are filled by host code with plain pointers and read by the
// kernels out of device memory; a pointer loaded from memory is a *flat* pointer to the compiler
// (flat_load: slower, and it ties vmcnt and lgkmcnt together).  In the device compilation pass the
// members are therefore declared in their real address space - global (1) for table / batch /
// output arrays, LDS (3) for the per-lane VM state - which turns every access into
// global_load / ds_read.  Same size and layout in both passes.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CBH_HOSTSIM)
#define CBH_G __attribute__((address_space(1)))
#define CBH_L __attribute__((address_space(3)))
#else
#define CBH_G
#define CBH_L
#endif

struct TableDev {
  const CBH_G u32* str_off; const CBH_G u8* str_bytes;
  const CBH_G u32* scope_parent; const CBH_G u32* scope_flags;
  const CBH_G CbhHashSlot* hash; u32 hash_mask;
  const CBH_G u32* rows; u32 n_rows;
  const CBH_G u32* rowleaf2;                                // [n_rows][8] fused-leaf records of the rows' derived-role conditions
  const CBH_G u32* drx;                                     // [n_dr][16] derived-role definitions for the flat kernel (CbhDrxField)
  const CBH_G u32* regex;                                   // CBH_SEC_REGEX: DFA tables of constant `matches` patterns
  const CBH_G u32* trace_rows; const CBH_G u32* trace_dr;   // CBH_SEC_TRACE_*: the trace pass's programs per rule / derived-role /
  const CBH_G u32* trace_rp; const CBH_G u32* trace_pool;   // role-policy record (cbh_trace_batch only)
  const CBH_G u32* rowpat;                                  // [n_rows][8] pattern halves (cbh_blob.h CbhRowPatField)
  const CBH_G u32* rowx; const CBH_G u32* rpx;              // CBH_SEC_ROWX / CBH_SEC_RPX (cbh_check_walk2.h)
  u32 gslots_generic, gslots_all;                           // evaluation-site slots of the table (CBH_M_GSLOTS_*)
  u32 inline_cols, sens_cols;                               // CBH_M_INLINE_COLS, CBH_M_SENS_COLS
  u32 q_sites;                                              // CBH_M_Q_SITES
  const CBH_G u8* str_wflags;                               // [K] CBH_SWF_* (CBH_SEC_STR_WFLAGS)
  const CBH_G u32* rprows; u32 n_rprows;
  const CBH_G u32* pool;
  const CBH_G u32* dr; u32 n_dr;
  const CBH_G u32* code;
  const CBH_G u8* const_tag; const CBH_G u64* const_val;
  const CBH_G u8* theap_tag; const CBH_G u64* theap_val;
  const CBH_G u32* const_rec; const CBH_G u32* theap_rec;   // the same values as 16-byte scalar-loadable records
  const CBH_G u8* role_class;                               // [K] role class of a string (cbh_blob.h CBH_SEC_ROLE_CLASS)
  const CBH_G u8* action_class;                             // [K] action class of a string (CBH_SEC_ACTION_CLASS)
  const CBH_G u64* gbits; u32 K;
  const CBH_G u64* nfa[3]; u32 nfa_words[3];
  u32 flags;
  u32 max_depth;                                            // longest scope chain of the table (entries), computed at load
  u32 n_scopes;                                             // scopes are numbered parents first (root = 0): a deeper scope has the larger index
  u32 max_bucket;                                           // most rule records in one resource-policy bucket, computed at load (selects cbh_check_flat_kernel_staged)
  const CBH_G u32* segs; const CBH_G u32* leafpool;         // CBH_SEC_SEGS / CBH_SEC_LEAFPOOL (the flat kernel's mask walk), null without
  u32 seg_info;                                             // CBH_M_SEGS
};

struct BatchDev {
  u32 n_requests, n_tuples, n_roles, n_columns, n_strings, heap_len;
  // the requests this launch decides: a chunk [req_lo, req_hi) of the batch (the arrays keep their
  // whole-batch layout, so a chunk's slices can be uploaded while another chunk is being decided)
  u32 req_lo, req_hi;
  const CBH_G u32* req_u32; const CBH_G u32* roles; const CBH_G u32* tuple_req; const CBH_G u32* tuple_action;
  const CBH_G u8* col_tag; const CBH_G u64* col_val;
  const CBH_G u8* heap_tag; const CBH_G u64* heap_val;
  const CBH_G u32* str_off; const CBH_G u8* str_bytes; const CBH_G u8* str_flags;
  // A batch the device flattened (cbh_wire.h): string i is the dictionary word str_keys[i] = {hash:16 | length:16 | offset:32},
  // its bytes str_bytes[offset ..) (the message buffer; an empty slot has length 0); str_off is null.  Else null.
  const CBH_G u64* str_keys;
  CBH_G u64* gbits; // [3][n_strings], written by the resolve kernel
  CBH_G u64* gres;  // [n_gwords][n_requests] results of the evaluation sites (cbh_walk2_pre_kernel writes, cbh_walk2_kernel reads)
  u32 n_gwords; u32 n_gslots;   // words per request; sites filed = slots 0 .. n_gslots - 1
  const CBH_G u32* ep_group;    // cbh_check_batch_trail: the group (Check call) request i belongs to, or null (one group)
  // the split pre-pass (cbh_walk2_collect_kernel / cbh_walk2_interp_kernel): per evaluation-site slot a list of (request | program << 32)
  CBH_G u32* site_cnt; CBH_G u64* site_list; u32 site_cap; u32 pad_sites;   // [slots], [slots][site_cap]; null = the fused pre-pass
};

struct OutDev {
  CBH_G u8* effect; CBH_G u32* policy; CBH_G u32* scope; CBH_G u8* status; CBH_G u64* edr;
  CBH_G u32* trace_rec; CBH_G u32* trace_cnt; u32 trace_cap; u32 ep_words;   // the trace pass's log (cerbos_hip.h cbh_trace), else null
  // cbh_check_batch_trail: [groups][ep_words] bit p = some binding of policy p (cbh_table_policy_key) was iterated for a request of
  // the group (check.go:302-304, the AuditTrail's effective policies); null = not wanted
  CBH_G u32* eff_pol;
};
// (idempotent: a stale read costs an atomic, never a bit.  The read goes to L2 - agent scope - on purpose: a CU's L1 would keep
// answering "not set" for the rest of the launch, and every later visit of every wave of that CU would queue an atomic on the one word)
__device__ __forceinline__ void ep_mark(const OutDev& o, const BatchDev& b, u32 req, u32 policy) {
  CBH_G u32* w = o.eff_pol + (size_t)(b.ep_group ? b.ep_group[req] : 0u) * o.ep_words + (policy >> 5);
  const u32 bit = 1u << (policy & 31u);
#ifdef CBH_HOSTSIM
  *w |= bit;
#else
  if (!(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr((unsigned int*)w, bit);
#endif
}

// Launch arguments of the decision kernel.  They live in device memory (one uniform pointer
// as the only kernel argument) so that every table / batch base address is a scalar load.
struct __attribute__((aligned(16))) KernelArgs { TableDev t; BatchDev b; OutDev o; long long now_ns; u32 flags; u32 pad; };

// Internal launch flags (KernelArgs.flags; never part of cbh_params.flags - the entry points mask them off): a batch whose
// requests mostly fit cbh_walk2_kernel's shape (<= 8 actions, <= 4 roles) is decided by it; the requests with more roles
// (<= 8) or more actions (<= 16) by the same walk in its wider forms (cbh_walk2_wide_kernel, cbh_walk2_awide_kernel: 64-bit
// walk vectors), and the few still wider ones by the general walk - each in a launch of its own over the same arrays, each kernel leaving the others' lanes alone.
#define CBH_FI_SKIP_WIDE 0x10000u   /* cbh_walk2_kernel / its pre-pass: a request wider than the base shape is not this launch's; */
                                    /* the wider walks / their pre-passes: only the requests of their class are (cbh_w2_class)   */
#define CBH_FI_ONLY_WIDE 0x20000u   /* cbh_check_kernel*: only the requests wider than the base shape are this launch's */
#define CBH_FI_ONLY_WIDER 0x40000u  /* cbh_check_kernel*: only the requests no shape of the walk holds are this launch's */
#define CBH_FI_PACKED_TAGS 0x80000u /* the column cache keeps a tag as a byte (CBH_CC_DWORDS): chosen per launch, where the smaller cache lets a CU hold more workgroups */
#define CBH_FI_MASK 0xF0000u
#define CBH_W2_NA 8u
#define CBH_W2_NR 4u
#define CBH_W2_WIDE_NR 8u           /* the wider shapes: 8 actions x 8 roles (cbh_walk2_wide_kernel) ... */
#define CBH_W2_AWIDE_NA 16u         /* ... and 16 actions x 4 roles (cbh_walk2_awide_kernel); both carry 64-bit walk vectors */
__device__ __forceinline__ bool cbh_is_wide(u32 act_cnt, u32 role_cnt) { return act_cnt > CBH_W2_NA || role_cnt > CBH_W2_NR; }
// which walk decides a request: 0 the base shape, 1 the shape with more roles, 2 the shape with more actions, 3 none of them (the general walk)
__device__ __forceinline__ u32 cbh_w2_class(u32 act_cnt, u32 role_cnt) {
  if (act_cnt <= CBH_W2_NA) return role_cnt <= CBH_W2_NR ? 0u : role_cnt <= CBH_W2_WIDE_NR ? 1u : 3u;
  return (act_cnt <= CBH_W2_AWIDE_NA && role_cnt <= CBH_W2_NR) ? 2u : 3u;
}
__device__ __forceinline__ bool cbh_is_wider(u32 act_cnt, u32 role_cnt) { return cbh_w2_class(act_cnt, role_cnt) == 3u; }

struct Val { u32 t; u64 v; };

struct Lane {           // per-lane evaluation state that programs can observe
  u32 req;
  u64 edr;              // effective derived roles of the scope being processed
  u32 status;           // CBH_ST_* accumulated
  bool edr_err;         // strict mode: derived roles of this scope failed to evaluate
  u32 pid;              // principal id, valid in the table walk only (leaf_fast reads P.id from here)
  u64 edr_errmask;      // trace pass only: WHICH derived roles failed (bit = name index), for the error's text
};

struct Ctx {
  const TableDev& t; const BatchDev& b;
  i64 now_ns; u32 flags; u32 tid;
  CBH_L u64* s_val; CBH_L u8* s_tag;       // operand stack   [CBH_STACK_DEPTH][CBH_BLOCK]
  CBH_L u64* l_val; CBH_L u8* l_tag;       // locals          [CBH_MAX_LOCALS][CBH_BLOCK]
  CBH_L u64* it_cont; CBH_L u32* it_idx; CBH_L u32* it_state; // iteration slots [CBH_MAX_ITERS][CBH_BLOCK]
  // Per-lane cache of the request's first n_cached attribute columns, filled once in the kernel
  // preamble (all loads in flight together) so condition leaves read LDS instead of paying one
  // HBM round trip each.  Three dword planes of [column][lane]: value low, value high, and the
  // aligned dword of col_tag that holds the lane's tag byte (the fill is an async global->LDS copy,
  // which moves dwords: cbh_check_wave.h).
  CBH_L u32* cc; u32 n_cached;
  // The launch arguments as they sit in device memory.  `t` / `b` above refer to a register copy
  // inside the kernels; functions that are real calls (slow paths, the stack interpreter) are
  // handed this pointer instead and build their own view, so the register copy's address never
  // escapes (it would be forced into scratch memory).
  const KernelArgs* ka_mem;
};
// the LDS part of a Ctx, passed by value across real calls
struct VmLds {
  CBH_L u64* s_val; CBH_L u8* s_tag; CBH_L u64* l_val; CBH_L u8* l_tag;
  CBH_L u64* it_cont; CBH_L u32* it_idx; CBH_L u32* it_state;
  CBH_L u32* cc; u32 n_cached; u32 tid;   // n_cached: bit 31 = the launch's CBH_FI_PACKED_TAGS (the arguments in memory are the batch's, not the launch's)
};
__device__ __forceinline__ VmLds lds_of(const Ctx& c) {
  return VmLds{c.s_val, c.s_tag, c.l_val, c.l_tag, c.it_cont, c.it_idx, c.it_state, c.cc, c.n_cached | ((c.flags & CBH_FI_PACKED_TAGS) ? 0x80000000u : 0u), c.tid};
}
__device__ __forceinline__ Ctx ctx_from_memory(const KernelArgs* ka, const VmLds& m) {
  return Ctx{ka->t, ka->b, ka->now_ns, (ka->flags & ~(u32)CBH_FI_PACKED_TAGS) | ((m.n_cached >> 31) ? CBH_FI_PACKED_TAGS : 0u), m.tid, m.s_val, m.s_tag, m.l_val, m.l_tag, m.it_cont, m.it_idx,
             m.it_state, m.cc, m.n_cached & 0x7FFFFFFFu, ka};
}
#define CBH_CACHE_COLS 16
// dwords of one wave's column cache (cbh_check_wave.h fill_column_cache): [value low word][value high word], each [column][lane],
// then the tags.  Two forms of those, chosen per launch (CBH_FI_PACKED_TAGS):
//   wide    [column][lane] the aligned dword of the batch's tag bytes that holds the lane's - every load of the fill goes straight
//           to LDS, nothing through a register: the form of a launch whose workgroups the cache does not limit (C2's 16 us kernel)
//   packed  [column / 4][lane][column % 4] a byte each: a quarter less LDS - one more workgroup to a CU for the nine to twelve
//           columns of C3, C4, T, C5 - for a load and a store through a register per column
#define CBH_CC_DWORDS(n, packed) ((2u * (n) + ((packed) ? ((n) + 3u) / 4u : (n))) * CBH_BLOCK)

__device__ __forceinline__ Val mk(u32 t, u64 v) { Val x; x.t = t; x.v = v; return x; }
__device__ __forceinline__ Val mk_err() { return mk(CBH_T_ERR, 0); }
// An error value's payload says which error it is (cerbos_hip.h CBH_ERR_*: code | detail << 8).  Only the trace pass reads
// it; an error that meets another keeps the payload of the one CEL would report (the left operand's).
__device__ __forceinline__ Val mk_errc(u32 code, u32 detail = 0) { return mk(CBH_T_ERR, (u64)code | ((u64)detail << 8)); }
__device__ __forceinline__ Val mk_bool(bool b) { return mk(CBH_T_BOOL, b ? 1u : 0u); }
__device__ __forceinline__ double as_f64(u64 v) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ u64 f64_bits(double d) { return (u64)__double_as_longlong(d); }
__device__ __forceinline__ bool is_num(u32 t) { return t == CBH_T_INT || t == CBH_T_UINT || t == CBH_T_DOUBLE; }

typedef const CBH_G u8* gbytes;   // string bytes live in the table image / batch pool (global memory)
// ---- strings -------------------------------------------------------------------------
__device__ __forceinline__ void str_span(const Ctx& c, u32 sid, gbytes& p, u32& n) {
  if (sid < c.t.K) {
    u32 o = c.t.str_off[sid];
    n = c.t.str_off[sid + 1] - o;
    p = c.t.str_bytes + o;
  } else {
    u32 i = sid - c.t.K;
    if (i >= c.b.n_strings) { n = 0; p = c.b.str_bytes; return; }   // never index past the batch's strings
    if (c.b.str_keys) { const u64 k = c.b.str_keys[i]; n = (u32)(k >> 32) & 0xFFFFu; p = c.b.str_bytes + (u32)k; return; }
    u32 o = c.b.str_off[i];
    n = c.b.str_off[i + 1] - o;
    p = c.b.str_bytes + o;
  }
}

__device__ inline int str_cmp(const Ctx& c, u32 a, u32 b) {
  if (a == b) return 0;
  gbytes pa, pb; u32 na, nb;
  str_span(c, a, pa, na); str_span(c, b, pb, nb);
  u32 n = na < nb ? na : nb;
  for (u32 i = 0; i < n; ++i) {
    if (pa[i] != pb[i]) return pa[i] < pb[i] ? -1 : 1;
  }
  return na < nb ? -1 : (na > nb ? 1 : 0);
}

// number of code points (CEL size(string))
__device__ inline u32 str_codepoints(const Ctx& c, u32 sid) {
  gbytes p; u32 n; str_span(c, sid, p, n);
  u32 k = 0;
  for (u32 i = 0; i < n; ++i) k += ((p[i] & 0xC0) != 0x80);
  return k;
}

__device__ inline bool str_find(const Ctx& c, u32 hay, u32 needle, int mode /*0 starts,1 ends,2 contains*/) {
  gbytes ph, pn; u32 nh, nn;
  str_span(c, hay, ph, nh); str_span(c, needle, pn, nn);
  if (nn > nh) return false;
  u32 lo = 0, hi = nh - nn;
  if (mode == 0) hi = 0;
  if (mode == 1) lo = hi;
  for (u32 s = lo; s <= hi; ++s) {
    u32 j = 0;
    while (j < nn && ph[s + j] == pn[j]) ++j;
    if (j == nn) return true;
  }
  return false;
}

// cel-go ext/strings.go indexOf / lastIndexOf (two-argument forms): index in CODE POINTS of the first / last occurrence of
// `sub` in `s`, -1 without one; the empty string is found at 0 / at the end.
__device__ inline i64 str_index_of(const Ctx& c, u32 s, u32 sub, bool last) {
  gbytes ph, pn; u32 nh, nn;
  str_span(c, s, ph, nh); str_span(c, sub, pn, nn);
  if (nn > nh) return -1;
  i64 at = -1;
  for (u32 o = 0; o + nn <= nh; ++o) {
    u32 j = 0;
    while (j < nn && ph[o + j] == pn[j]) ++j;
    if (j == nn) { at = o; if (!last) break; }
  }
  if (at < 0) return -1;
  i64 cp = 0;
  for (u32 i = 0; i < (u32)at; ++i) cp += (ph[i] & 0xC0u) != 0x80u;   // bytes that start a code point
  return cp;
}
// a == b through ASCII case mappings (mode 0 as is, 1 lowerAscii, 2 upperAscii), byte by byte: no string is built
__device__ inline bool str_eq_case(const Ctx& c, u32 a, u32 ma, u32 b, u32 mb) {
  gbytes pa, pb; u32 na, nb;
  str_span(c, a, pa, na); str_span(c, b, pb, nb);
  if (na != nb) return false;
  auto map = [](u8 ch, u32 m) -> u8 {
    if (m == 1 && ch >= 'A' && ch <= 'Z') return (u8)(ch + 32);
    if (m == 2 && ch >= 'a' && ch <= 'z') return (u8)(ch - 32);
    return ch;
  };
  for (u32 i = 0; i < na; ++i) if (map(pa[i], ma) != map(pb[i], mb)) return false;
  return true;
}

// RE2 MatchString with the pattern's DFA (cerbos_amd/lower/regex.py, layout in cbh_blob.h CBH_SEC_REGEX): one table
// lookup per byte; flags bit 0 = a match is already certain, bit 1 = a match if the text ends in this state.
__device__ inline bool regex_match(const Ctx& c, u32 off, u32 sid) {
  const CBH_G u32* r = c.t.regex + off;
  const u32 n_states = r[0], n_cls = r[1];
  const CBH_G u32* classmap = r + 2;
  const CBH_G u32* flags = r + 66;
  const CBH_G u32* trans = flags + n_states;
  gbytes p; u32 n;
  str_span(c, sid, p, n);
  u32 s = 0;
  if (flags[0] & 1u) return true;
  for (u32 i = 0; i < n; ++i) {
    const u32 b = p[i];
    const u32 cls = (classmap[b >> 2] >> ((b & 3u) * 8u)) & 0xFFu;
    s = trans[s * n_cls + cls];
    if (flags[s] & 1u) return true;
  }
  return (flags[s] & 2u) != 0;
}

// cerbos.lib.hierarchy predicates (internal/conditions/types/hierarchy.go:259-385) on the two dot-delimited strings
// themselves: strings.Split(s, ".") never yields a segment with a dot in it, so "the segments of P are the first
// segments of Q" is "Q starts with P's bytes and continues with '.' or ends there", and segment counts are dot counts.
__device__ inline bool hier_pred(const Ctx& c, u32 kind, u32 a, u32 b) {
  gbytes pa, pb; u32 na, nb;
  str_span(c, a, pa, na); str_span(c, b, pb, nb);
  if (kind == 1 || kind == 3) { gbytes tp = pa; pa = pb; pb = tp; const u32 tn = na; na = nb; nb = tn; kind -= 1; }   // descendentOf / immediateChildOf: the mirrored question
  auto dots = [](gbytes p, u32 from, u32 n) { u32 k = 0; for (u32 i = from; i < n; ++i) k += p[i] == '.'; return k; };
  auto seg_prefix = [](gbytes p, u32 np, gbytes q, u32 nq) {   // P's segments lead Q's
    if (nq < np) return false;
    for (u32 i = 0; i < np; ++i) if (p[i] != q[i]) return false;
    return nq == np || q[np] == '.';
  };
  if (kind == 0) return nb > na && seg_prefix(pa, na, pb, nb);                                   // ancestorOf: strictly more segments
  if (kind == 2) return nb > na && seg_prefix(pa, na, pb, nb) && dots(pb, na + 1, nb) == 0;      // immediateParentOf: exactly one more
  const u32 da = dots(pa, 0, na), db = dots(pb, 0, nb);
  if (kind == 4) {   // siblingOf: as many segments, all but the last equal
    if (da != db) return false;
    u32 la = na, lb = nb;   // one past the last dot, 0 without one
    while (la > 0 && pa[la - 1] != '.') --la;
    while (lb > 0 && pb[lb - 1] != '.') --lb;
    if (la != lb) return false;
    for (u32 i = 0; i < la; ++i) if (pa[i] != pb[i]) return false;
    return true;
  }
  return da <= db ? seg_prefix(pa, na, pb, nb) : seg_prefix(pb, nb, pa, na);   // overlaps: the shorter leads the longer
}

// hierarchy(a).commonAncestors(hierarchy(b)) == hierarchy(c) over the three dot-delimited strings (types/hierarchy.go:296-322:
// hierarchies of equal length are both cut by their last segment first; the ancestors are the leading segments they share).
// An empty ancestor list equals no hierarchy a string spells (strings.Split never returns an empty slice).
__device__ inline bool hier_common_eq(const Ctx& c, u32 a, u32 b, u32 cc) {
  gbytes pa, pb, pc; u32 na, nb, nc;
  str_span(c, a, pa, na); str_span(c, b, pb, nb); str_span(c, cc, pc, nc);
  u32 sa = 1, sb = 1;
  for (u32 i = 0; i < na; ++i) sa += pa[i] == '.';
  for (u32 i = 0; i < nb; ++i) sb += pb[i] == '.';
  const u32 limit = sa == sb ? sa - 1u : (sa < sb ? sa : sb);
  u32 ia = 0, ib = 0, m = 0, end_a = 0;   // end_a: one past the m-th shared segment of a
  while (m < limit) {
    u32 ea = ia, eb = ib;
    while (ea < na && pa[ea] != '.') ++ea;
    while (eb < nb && pb[eb] != '.') ++eb;
    bool same = (ea - ia) == (eb - ib);
    for (u32 k = 0; same && k < ea - ia; ++k) same = pa[ia + k] == pb[ib + k];
    if (!same) break;
    ++m; end_a = ea; ia = ea + 1; ib = eb + 1;
  }
  if (m == 0 || end_a != nc) return false;
  for (u32 i = 0; i < nc; ++i) if (pa[i] != pc[i]) return false;
  return true;
}

// ---- heap ----------------------------------------------------------------------------
__device__ __forceinline__ u32 cont_sel(u64 v) { return (u32)(v >> 62); }
__device__ __forceinline__ u32 cont_off(u64 v) { return (u32)((v >> 32) & 0x3FFFFFFFu); }
__device__ __forceinline__ u32 cont_len(u64 v) { return (u32)v; }

// The arena of a lane: CBH_ARENA_ENTRIES values [slot][lane] in dynamic LDS behind the column cache (the launch sizes it when the
// table has CBH_MF_NEEDS_ARENA).  Lists a program builds live there for the program's duration (cbh_interp.h bumps a pointer).
__device__ __forceinline__ CBH_L u64* arena_vals(const Ctx& c) { return (CBH_L u64*)(c.cc + CBH_CC_DWORDS(c.n_cached, (c.flags & CBH_FI_PACKED_TAGS) != 0)); }
__device__ __forceinline__ CBH_L u8* arena_tags(const Ctx& c) { return (CBH_L u8*)(arena_vals(c) + CBH_ARENA_ENTRIES * CBH_BLOCK); }
__device__ __forceinline__ void arena_put(const Ctx& c, u32 idx, Val v) {
  arena_vals(c)[idx * CBH_BLOCK + c.tid] = v.v; arena_tags(c)[idx * CBH_BLOCK + c.tid] = (u8)v.t;
}

__device__ __forceinline__ Val heap_get(const Ctx& c, u32 sel, u32 idx) {
  if (sel == CBH_HEAP_TABLE) return mk(c.t.theap_tag[idx], c.t.theap_val[idx]);
  if (sel == CBH_HEAP_BATCH) return mk(c.b.heap_tag[idx], c.b.heap_val[idx]);
  if (sel == CBH_HEAP_LOCAL) return idx < CBH_ARENA_ENTRIES ? mk(arena_tags(c)[idx * CBH_BLOCK + c.tid], arena_vals(c)[idx * CBH_BLOCK + c.tid]) : mk_err();
  return mk(CBH_T_STRING, c.b.roles[idx]);
}

// ---- ropes: strings a program puts together are never built -----------------------------------
// A rope is the list of its parts in the lane's arena - string id | case mode << 32 (1 lowerAscii, 2 upperAscii: the part is read
// through the mapping) -; whoever needs its bytes walks the parts.  Only the operand-stack interpreter makes and reads ropes
// (cbh_interp.h: equality, `in`, prefix / suffix / substring search, size(); anything else flags the tuple UNSUPPORTED), through
// ONE out-of-line function (rope_op below), so that the shared comparison code and the kernels without an interpreter stay as they are.
// A rope never becomes a list element.
__device__ __forceinline__ u32 rope_parts(u64 v) { return (u32)v & 0xFFFFu; }
__device__ __forceinline__ Val mk_rope(u32 off, u32 parts) { return mk(CBH_T_ROPE, ((u64)CBH_HEAP_LOCAL << 62) | ((u64)off << 32) | parts); }
__device__ __forceinline__ bool is_strlike(u32 t) { return t == CBH_T_STRING || t == CBH_T_ROPE; }
// A part may be a WINDOW of its string (substring / charAt / trim / the pieces replace() keeps): bits 34-48 = first byte,
// bits 49-63 = bytes + 1 (0 = the whole string).
#define CBH_ROPE_WINDOW_MAX 0x7FFEu
__device__ __forceinline__ u64 rope_window(u32 sid, u32 first, u32 bytes) { return (u64)sid | ((u64)first << 34) | ((u64)(bytes + 1u) << 49); }
__device__ __forceinline__ void part_span(const Ctx& c, u64 part, gbytes& p, u32& n) {
  str_span(c, (u32)part, p, n);
  const u32 ln = (u32)(part >> 49) & 0x7FFFu;
  if (ln) { p += (u32)(part >> 34) & 0x7FFFu; n = ln - 1u; }
}
__device__ inline u32 sl_len(const Ctx& c, Val x) {   // bytes of a string or rope
  gbytes p; u32 n;
  if (x.t == CBH_T_STRING) { str_span(c, (u32)x.v, p, n); return n; }
  u32 tot = 0;
  for (u32 k = 0; k < rope_parts(x.v); ++k) { part_span(c, heap_get(c, CBH_HEAP_LOCAL, cont_off(x.v) + k).v, p, n); tot += n; }
  return tot;
}
__device__ inline u32 sl_byte(const Ctx& c, Val x, u32 i) {   // its i-th byte, through the part's case mode (i < sl_len)
  gbytes p; u32 n; u32 b = 0;
  if (x.t == CBH_T_STRING) { str_span(c, (u32)x.v, p, n); return i < n ? p[i] : 0u; }
  u32 mode = 0;
  for (u32 k = 0; k < rope_parts(x.v); ++k) {
    const u64 part = heap_get(c, CBH_HEAP_LOCAL, cont_off(x.v) + k).v;
    part_span(c, part, p, n);
    if (i < n) { b = p[i]; mode = (u32)(part >> 32) & 3u; break; }
    i -= n;
  }
  if (mode == 1 && b >= 'A' && b <= 'Z') b += 32;
  if (mode == 2 && b >= 'a' && b <= 'z') b -= 32;
  return b;
}
__device__ inline bool sl_equal(const Ctx& c, Val a, Val b) {
  if (!is_strlike(a.t) || !is_strlike(b.t)) return false;   // a rope equals nothing that is not a string
  const u32 n = sl_len(c, a);
  if (n != sl_len(c, b)) return false;
  for (u32 i = 0; i < n; ++i) if (sl_byte(c, a, i) != sl_byte(c, b, i)) return false;
  return true;
}

// ---- numeric comparison (exact across int64 / uint64 / double) -------------------------
// returns -1, 0, 1 or 2 (unordered: NaN)
__device__ inline int cmp_i64_f64(i64 i, double d) {
  if (d != d) return 2;
  if (d >= 9223372036854775808.0) return -1;
  if (d < -9223372036854775808.0) return 1;
  double tr = trunc(d);
  i64 t = (i64)tr;
  if (i != t) return i < t ? -1 : 1;
  double fr = d - tr;
  return fr > 0 ? -1 : (fr < 0 ? 1 : 0);
}
__device__ inline int cmp_u64_f64(u64 u, double d) {
  if (d != d) return 2;
  if (d >= 18446744073709551616.0) return -1;
  if (d < 0) return 1;
  double tr = trunc(d);
  u64 t = (u64)tr;
  if (u != t) return u < t ? -1 : 1;
  double fr = d - tr;
  return fr > 0 ? -1 : (fr < 0 ? 1 : 0);
}
__device__ inline int num_cmp(Val a, Val b) {
  if (a.t == CBH_T_DOUBLE && b.t == CBH_T_DOUBLE) {
    double x = as_f64(a.v), y = as_f64(b.v);
    if (x != x || y != y) return 2;
    return x < y ? -1 : (x > y ? 1 : 0);
  }
  if (a.t == CBH_T_INT) {
    i64 x = (i64)a.v;
    if (b.t == CBH_T_INT) { i64 y = (i64)b.v; return x < y ? -1 : (x > y ? 1 : 0); }
    if (b.t == CBH_T_UINT) { if (x < 0) return -1; u64 ux = (u64)x; return ux < b.v ? -1 : (ux > b.v ? 1 : 0); }
    return cmp_i64_f64(x, as_f64(b.v));
  }
  if (a.t == CBH_T_UINT) {
    if (b.t == CBH_T_UINT) return a.v < b.v ? -1 : (a.v > b.v ? 1 : 0);
    if (b.t == CBH_T_INT) { i64 y = (i64)b.v; if (y < 0) return 1; return a.v < (u64)y ? -1 : (a.v > (u64)y ? 1 : 0); }
    return cmp_u64_f64(a.v, as_f64(b.v));
  }
  // a double, b int/uint
  int r = (b.t == CBH_T_INT) ? cmp_i64_f64((i64)b.v, as_f64(a.v)) : cmp_u64_f64(b.v, as_f64(a.v));
  return r == 2 ? 2 : -r;
}

// scalar equality (cel-go Equal): mismatched types are simply unequal.
// `deep` receives true when a container comparison needs more than one nesting level.
__device__ inline bool scalar_equal(const Ctx& c, Val a, Val b) {
  if (is_num(a.t) && is_num(b.t)) return num_cmp(a, b) == 0;
  if (a.t != b.t) return false;
  switch (a.t) {
    case CBH_T_NULL: return true;
    case CBH_T_BOOL: case CBH_T_STRING: case CBH_T_TIMESTAMP: case CBH_T_DURATION: return a.v == b.v;
    default: return false;
  }
}

__device__ inline bool val_equal(const Ctx& c, Lane& L, Val a, Val b) {
  if (a.t == CBH_T_LIST && b.t == CBH_T_LIST) {
    u32 n = cont_len(a.v);
    if (n != cont_len(b.v)) return false;
    for (u32 i = 0; i < n; ++i) {
      Val x = heap_get(c, cont_sel(a.v), cont_off(a.v) + i);
      Val y = heap_get(c, cont_sel(b.v), cont_off(b.v) + i);
      if (x.t == CBH_T_LIST || x.t == CBH_T_MAP || y.t == CBH_T_LIST || y.t == CBH_T_MAP) {
        L.status |= CBH_ST_UNSUPPORTED;  // nested container equality is not on the device
        return false;
      }
      if (!scalar_equal(c, x, y)) return false;
    }
    return true;
  }
  if (a.t == CBH_T_MAP && b.t == CBH_T_MAP) { L.status |= CBH_ST_UNSUPPORTED; return false; }
  return scalar_equal(c, a, b);
}

// ordering: returns -1/0/1, 2 = unordered (NaN -> every comparison false), 3 = no such overload
__device__ inline int val_compare(const Ctx& c, Val a, Val b) {
  if (is_num(a.t) && is_num(b.t)) return num_cmp(a, b);
  if (a.t != b.t) return 3;
  switch (a.t) {
    case CBH_T_BOOL: return a.v < b.v ? -1 : (a.v > b.v ? 1 : 0);
    case CBH_T_STRING: return str_cmp(c, (u32)a.v, (u32)b.v);
    case CBH_T_TIMESTAMP: case CBH_T_DURATION: { i64 x = (i64)a.v, y = (i64)b.v; return x < y ? -1 : (x > y ? 1 : 0); }
    default: return 3;
  }
}

// map lookup by string key: entries are (key, value) pairs
__device__ inline bool map_find(const Ctx& c, Val m, Val key, Val& out) {
  u32 n = cont_len(m.v), off = cont_off(m.v), sel = cont_sel(m.v);
  for (u32 i = 0; i < n; ++i) {
    Val k = heap_get(c, sel, off + 2 * i);
    if (scalar_equal(c, k, key)) { out = heap_get(c, sel, off + 2 * i + 1); return true; }
  }
  return false;
}

// ---- time ----------------------------------------------------------------------------
__device__ inline i64 days_from_civil(i64 y, u32 m, u32 d) {
  y -= m <= 2;
  const i64 era = (y >= 0 ? y : y - 399) / 400;
  const u32 yoe = (u32)(y - era * 400);
  const u32 doy = (153 * (m > 2 ? m - 3 : m + 9) + 2) / 5 + d - 1;
  const u32 doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + (i64)doe - 719468;
}
// Getters of cel-go's timestamp.go / duration.go.  kind: 0 getFullYear, 1 getMonth (0-based), 2 getDayOfYear (0-based),
// 3 getDayOfMonth (0-based), 4 getDate, 5 getDayOfWeek (Sunday = 0), 6 getHours, 7 getMinutes, 8 getSeconds,
// 9 getMilliseconds.  A timestamp is read in the fixed zone `off_s` seconds east of UTC; a duration has the last four
// only: the whole duration in that unit, truncated towards zero (Go's float conversion of d.Hours() etc.).
__device__ inline bool ts_getter(Val x, u32 kind, i64 off_s, i64& out) {
  if (x.t == CBH_T_DURATION) {
    const i64 d = (i64)x.v;
    switch (kind) {
      case 6: out = d / 3600000000000ll; return true;
      case 7: out = d / 60000000000ll; return true;
      case 8: out = d / 1000000000ll; return true;
      case 9: out = d / 1000000ll; return true;
      default: return false;   // no such overload
    }
  }
  if (x.t != CBH_T_TIMESTAMP || kind > 9) return false;
  i64 t;
  if (__builtin_add_overflow((i64)x.v, off_s * 1000000000ll, &t)) return false;
  i64 secs = t / 1000000000ll, nsec = t % 1000000000ll;
  if (nsec < 0) { nsec += 1000000000ll; --secs; }
  i64 days = secs / 86400, sod = secs % 86400;
  if (sod < 0) { sod += 86400; --days; }
  if (kind >= 6) {
    out = kind == 6 ? sod / 3600 : kind == 7 ? (sod / 60) % 60 : kind == 8 ? sod % 60 : nsec / 1000000ll;
    return true;
  }
  if (kind == 5) { i64 w = (days + 4) % 7; out = w < 0 ? w + 7 : w; return true; }   // 1970-01-01 was a Thursday
  // civil date from days since the epoch (the inverse of days_from_civil)
  const i64 z = days + 719468;
  const i64 era = (z >= 0 ? z : z - 146096) / 146097;
  const u32 doe = (u32)(z - era * 146097);
  const u32 yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const u32 doy = doe - (365 * yoe + yoe / 4 - yoe / 100);   // from March 1st
  const u32 mp = (5 * doy + 2) / 153;
  const u32 d = doy - (153 * mp + 2) / 5 + 1;
  const u32 m = mp < 10 ? mp + 3 : mp - 9;
  const i64 y = (i64)yoe + era * 400 + (m <= 2);
  switch (kind) {
    case 0: out = y; break;
    case 1: out = (i64)m - 1; break;
    case 2: out = days - days_from_civil(y, 1, 1); break;
    case 3: out = (i64)d - 1; break;
    default: out = d; break;   // 4 getDate
  }
  return true;
}
__device__ __forceinline__ bool dig(u8 ch) { return ch >= '0' && ch <= '9'; }

// RFC 3339 (Go time.Parse(time.RFC3339)): returns 0 ok, 1 parse error, 2 outside the i64-ns range
__device__ inline int parse_timestamp(gbytes p, u32 n, i64& out_ns) {
  if (n < 20) return 1;
  for (int i = 0; i < 19; ++i) {
    bool d = dig(p[i]);
    if (i == 4 || i == 7) { if (p[i] != '-') return 1; }
    else if (i == 10) { if (p[i] != 'T') return 1; }
    else if (i == 13 || i == 16) { if (p[i] != ':') return 1; }
    else if (!d) return 1;
  }
  u32 Y = (p[0] - '0') * 1000 + (p[1] - '0') * 100 + (p[2] - '0') * 10 + (p[3] - '0');
  u32 M = (p[5] - '0') * 10 + (p[6] - '0'), D = (p[8] - '0') * 10 + (p[9] - '0');
  u32 h = (p[11] - '0') * 10 + (p[12] - '0'), mi = (p[14] - '0') * 10 + (p[15] - '0'), s = (p[17] - '0') * 10 + (p[18] - '0');
  if (M < 1 || M > 12 || D < 1 || h > 23 || mi > 59 || s > 59) return 1;
  const u32 mdays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  u32 dim = mdays[M - 1];
  if (M == 2 && ((Y % 4 == 0 && Y % 100 != 0) || Y % 400 == 0)) dim = 29;
  if (D > dim) return 1;
  u32 i = 19; i64 frac = 0;
  if (p[i] == '.') {
    ++i; u32 nd = 0;
    while (i < n && dig(p[i])) { if (nd < 9) { frac = frac * 10 + (p[i] - '0'); ++nd; } ++i; }
    if (nd == 0 && !(i > 20)) return 1;
    while (nd < 9) { frac *= 10; ++nd; }
  }
  if (i >= n) return 1;
  i64 off = 0;
  if (p[i] == 'Z') { if (i + 1 != n) return 1; }
  else if (p[i] == '+' || p[i] == '-') {
    if (i + 6 != n || !dig(p[i + 1]) || !dig(p[i + 2]) || p[i + 3] != ':' || !dig(p[i + 4]) || !dig(p[i + 5])) return 1;
    u32 oh = (p[i + 1] - '0') * 10 + (p[i + 2] - '0'), om = (p[i + 4] - '0') * 10 + (p[i + 5] - '0');
    if (oh > 23 || om > 59) return 1;
    off = (i64)(oh * 3600 + om * 60); if (p[i] == '-') off = -off;
  } else return 1;
  i64 secs = days_from_civil(Y, M, D) * 86400 + (i64)(h * 3600 + mi * 60 + s) - off;
  if (secs < -62135596800LL || secs > 253402300799LL) return 1;   // CEL timestamp range
  if (secs < -9223372036LL || secs > 9223372035LL) return 2;      // not representable in i64 ns
  out_ns = secs * 1000000000LL + frac;
  return 0;
}

// Go time.ParseDuration: returns 0 ok, 1 error
__device__ inline int parse_duration(gbytes p, u32 n, i64& out_ns) {
  u32 i = 0; bool neg = false;
  if (n == 0) return 1;
  if (p[0] == '-' || p[0] == '+') { neg = p[0] == '-'; i = 1; }
  if (i == n) return 1;
  if (n - i == 1 && p[i] == '0') { out_ns = 0; return 0; }
  u64 d = 0;
  while (i < n) {
    u64 v = 0, f = 0; double scale = 1; bool pre = false, post = false;
    if (!(p[i] == '.' || dig(p[i]))) return 1;
    u32 pl = i;
    while (i < n && dig(p[i])) {
      if (v > ((1ull << 63) - 1) / 10) return 1;
      v = v * 10 + (p[i] - '0');
      if (v > (1ull << 63)) return 1;
      ++i;
    }
    pre = pl != i;
    if (i < n && p[i] == '.') {
      ++i; u32 pl2 = i; bool overflow = false;
      while (i < n && dig(p[i])) {
        if (!overflow) {
          if (f > ((1ull << 63) - 1) / 10) overflow = true;
          else {
            u64 y = f * 10 + (p[i] - '0');
            if (y > (1ull << 63)) overflow = true; else { f = y; scale *= 10; }
          }
        }
        ++i;
      }
      post = pl2 != i;
    }
    if (!pre && !post) return 1;
    u32 us = i;
    while (i < n && p[i] != '.' && !dig(p[i])) ++i;
    u32 ul = i - us; if (ul == 0) return 1;
    u64 unit = 0; gbytes q = p + us;
    if (ul == 2 && q[0] == 'n' && q[1] == 's') unit = 1;
    else if (ul == 2 && q[0] == 'u' && q[1] == 's') unit = 1000;
    else if (ul == 3 && q[0] == 0xC2 && q[1] == 0xB5 && q[2] == 's') unit = 1000;
    else if (ul == 3 && q[0] == 0xCE && q[1] == 0xBC && q[2] == 's') unit = 1000;
    else if (ul == 2 && q[0] == 'm' && q[1] == 's') unit = 1000000;
    else if (ul == 1 && q[0] == 's') unit = 1000000000ull;
    else if (ul == 1 && q[0] == 'm') unit = 60000000000ull;
    else if (ul == 1 && q[0] == 'h') unit = 3600000000000ull;
    else return 1;
    if (v > (1ull << 63) / unit) return 1;
    v *= unit;
    if (f > 0) {
      v += (u64)((double)f * ((double)unit / scale));
      if (v > (1ull << 63)) return 1;
    }
    d += v;
    if (d > (1ull << 63)) return 1;
  }
  if (neg) { out_ns = (i64)(0 - d); return 0; }
  if (d > (1ull << 63) - 1) return 1;
  out_ns = (i64)d;
  return 0;
}

// dotted IPv4 "a.b.c.d" (Go netip / net.ParseIP accept no leading zeros > 1 digit... net.ParseCIDR and
// net.ParseIP reject leading zeros since Go 1.17): returns false on any deviation.
__device__ inline bool parse_ipv4(gbytes p, u32 n, u32& out) {
  u32 v = 0, parts = 0, i = 0;
  while (parts < 4) {
    if (i >= n || !dig(p[i])) return false;
    u32 x = 0, nd = 0, st = i;
    while (i < n && dig(p[i])) { x = x * 10 + (p[i] - '0'); ++nd; ++i; if (x > 255) return false; }
    if (nd > 1 && p[st] == '0') return false;
    v = (v << 8) | x; ++parts;
    if (parts < 4) { if (i >= n || p[i] != '.') return false; ++i; }
  }
  if (i != n) return false;
  out = v; return true;
}

// IPv6 text as Go's net.ParseIP (netip.ParseAddr) reads it: up to eight groups of at most four hex digits, one "::" that
// stands for at least one zero group, optionally a dotted IPv4 in the last 32 bits; no zone.  -> 128 bits (hi, lo).
__device__ inline bool parse_ipv6(gbytes p, u32 n, u64& hi, u64& lo) {
  u32 g[8]; int ng = 0, ellipsis = -1; u32 i = 0;
  for (int k = 0; k < 8; ++k) g[k] = 0;
  if (n >= 2 && p[0] == ':' && p[1] == ':') { ellipsis = 0; i = 2; }
  else if (n == 0 || p[0] == ':') return false;
  while (i < n) {
    if (ng >= 8) return false;
    u32 v = 0, nd = 0; const u32 st = i;
    for (; i < n; ++i) {
      const u8 ch = p[i];
      u32 d;
      if (ch >= '0' && ch <= '9') d = ch - '0'; else if (ch >= 'a' && ch <= 'f') d = ch - 'a' + 10; else if (ch >= 'A' && ch <= 'F') d = ch - 'A' + 10; else break;
      if (++nd > 4) return false;
      v = (v << 4) | d;
    }
    if (nd == 0) return false;
    if (i < n && p[i] == '.') {   // the rest is a dotted IPv4: the last two groups
      if (ng > 6) return false;
      u32 v4;
      if (!parse_ipv4(p + st, n - st, v4)) return false;
      g[ng++] = v4 >> 16; g[ng++] = v4 & 0xFFFFu;
      i = n;
      break;
    }
    g[ng++] = v;
    if (i == n) break;
    if (p[i] != ':') return false;
    if (++i == n) return false;                 // a single trailing colon
    if (p[i] == ':') {
      if (ellipsis >= 0) return false;          // a second "::"
      ellipsis = ng;
      if (++i == n) break;
    }
  }
  if (ng < 8) {
    if (ellipsis < 0) return false;
    const int tail = ng - ellipsis;             // groups behind the "::" move to the end
    for (int k = tail - 1; k >= 0; --k) g[8 - tail + k] = g[ellipsis + k];
    for (int k = ellipsis; k < 8 - tail; ++k) g[k] = 0;
  } else if (ellipsis >= 0) return false;       // "::" must stand for at least one group
  hi = ((u64)g[0] << 48) | ((u64)g[1] << 32) | ((u64)g[2] << 16) | (u64)g[3];
  lo = ((u64)g[4] << 48) | ((u64)g[5] << 32) | ((u64)g[6] << 16) | (u64)g[7];
  return true;
}

// inIPAddrRange (internal/conditions/cerbos_lib.go:513-526: net.ParseIP + net.ParseCIDR + IPNet.Contains).
// 1 / 0 = contained / not, -1 = one of them does not parse (an error in the reference), -2 = a form the device leaves to the
// caller's engine (an IPv4-mapped IPv6 network, whose mask Go re-reads as an IPv4 mask).
__device__ inline int ip_in_range(gbytes pi, u32 ni, gbytes pc, u32 nc) {
  u32 slash = nc;
  for (u32 i = 0; i < nc; ++i) if (pc[i] == '/') { slash = i; break; }
  if (slash >= nc) return -1;
  u32 bits = 0, nd = 0;
  for (u32 i = slash + 1; i < nc; ++i) { if (!dig(pc[i]) || ++nd > 6) return -1; bits = bits * 10 + (pc[i] - '0'); }   // Go's dtoi: plain digits
  if (nd == 0) return -1;
  bool ip6 = false, net6 = false;
  for (u32 i = 0; i < ni; ++i) ip6 |= pi[i] == ':';
  for (u32 i = 0; i < slash; ++i) net6 |= pc[i] == ':';
  u64 ih = 0, il = 0, nh = 0, nl = 0; u32 v4;
  if (ip6) { if (!parse_ipv6(pi, ni, ih, il)) return -1; }
  else { if (!parse_ipv4(pi, ni, v4)) return -1; il = v4; }
  if (net6) { if (!parse_ipv6(pc, slash, nh, nl)) return -1; }
  else { if (!parse_ipv4(pc, slash, v4)) return -1; nl = v4; }
  if (bits > (net6 ? 128u : 32u)) return -1;
  if (net6 && nh == 0 && (nl >> 32) == 0xFFFFull) return -2;
  if (ip6 && ih == 0 && (il >> 32) == 0xFFFFull) { ip6 = false; il &= 0xFFFFFFFFull; }   // IP.To4(): an IPv4-mapped address is an IPv4 address
  if (ip6 != net6) return 0;                                                                 // different lengths: not contained
  if (!net6) {
    const u32 mask = bits == 0 ? 0u : (0xFFFFFFFFu << (32 - bits));
    return (((u32)il ^ (u32)nl) & mask) == 0;
  }
  const u64 mh = bits == 0 ? 0ull : (bits >= 64 ? ~0ull : (~0ull << (64 - bits)));
  const u64 ml = bits <= 64 ? 0ull : (bits >= 128 ? ~0ull : (~0ull << (128 - bits)));
  return ((ih ^ nh) & mh) == 0 && ((il ^ nl) & ml) == 0;
}

// netip.ParseAddr as cel-go's ext.Network reads an address (isIP / ip() / cidr().containsIP): no zone, and an IPv4-mapped
// IPv6 address is refused.  -> 0 not an address, else the family (4 / 6) and the 128 bits.
__device__ inline u32 parse_addr(gbytes p, u32 n, u64& hi, u64& lo) {
  bool v6 = false;
  for (u32 i = 0; i < n; ++i) v6 |= p[i] == ':';
  hi = lo = 0;
  if (!v6) { u32 v4; if (!parse_ipv4(p, n, v4)) return 0; lo = v4; return 4; }
  if (!parse_ipv6(p, n, hi, lo)) return 0;
  if (hi == 0 && (lo >> 32) == 0xFFFFull) return 0;
  return 6;
}
// the predicates of netip.Addr (family 4: the address in the low 32 bits of `lo`)
__device__ inline bool addr_is(u32 what, u32 fam, u64 hi, u64 lo) {
  const bool v4 = fam == 4;
  const u32 a4 = (u32)lo;
  const bool unspec = v4 ? a4 == 0 : (hi == 0 && lo == 0);
  const bool loop = v4 ? (a4 >> 24) == 127u : (hi == 0 && lo == 1);
  const bool multicast = v4 ? (a4 >> 28) == 0xEu : (hi >> 56) == 0xFFull;
  const bool ll_uni = v4 ? (a4 >> 16) == 0xA9FEu : (hi >> 54) == (0xFE80ull >> 6);
  const bool ll_multi = v4 ? (a4 >> 8) == 0xE00000u : ((hi >> 48) & 0xFF0Full) == 0xFF02ull;
  switch (what) {
    case 2: return unspec;
    case 3: return loop;
    case 4: return ll_uni;
    case 5: return ll_multi;
    default: return !unspec && !(v4 && a4 == 0xFFFFFFFFu) && !loop && !multicast && !ll_uni;   // 6: isGlobalUnicast
  }
}
// Is the text netip.Addr.String() of the address it spells?  (IPv4: the parser admits nothing else; IPv6: RFC 5952 -
// lower-case hex without leading zeros, the longest run of two or more zero groups, leftmost on a tie, as "::")
__device__ inline bool addr_is_canonical(gbytes p, u32 n, u32 fam, u64 hi, u64 lo) {
  if (fam == 4) return true;
  u32 g[8];
  for (int k = 0; k < 4; ++k) { g[k] = (u32)(hi >> (48 - 16 * k)) & 0xFFFFu; g[4 + k] = (u32)(lo >> (48 - 16 * k)) & 0xFFFFu; }
  int best = -1, best_len = 0;
  for (int k = 0; k < 8;) {
    if (g[k] != 0) { ++k; continue; }
    int e = k;
    while (e < 8 && g[e] == 0) ++e;
    if (e - k >= 2 && e - k > best_len) { best = k; best_len = e - k; }
    k = e;
  }
  u32 i = 0;
  for (int k = 0; k < 8; ++k) {
    if (k == best) {
      if (i + 2 > n || p[i] != ':' || p[i + 1] != ':') return false;
      i += 2; k += best_len - 1;
      continue;
    }
    if (k > 0 && k != best + best_len) { if (i >= n || p[i] != ':') return false; ++i; }
    bool started = false;
    for (int sh = 12; sh >= 0; sh -= 4) {
      const u32 d = (g[k] >> sh) & 0xFu;
      if (d == 0 && !started && sh != 0) continue;
      started = true;
      const u8 ch = (u8)(d < 10 ? '0' + d : 'a' + d - 10);
      if (i >= n || p[i] != ch) return false;
      ++i;
    }
  }
  return i == n;
}

// ---- interpreter -----------------------------------------------------------------------
#define ST(i) c.s_tag[(i) * CBH_BLOCK + c.tid]
#define SV(i) c.s_val[(i) * CBH_BLOCK + c.tid]
#define PUSHV(x) do { Val _x = (x); ST(sp) = (u8)_x.t; SV(sp) = _x.v; ++sp; } while (0)
#define TOPV(k) mk(ST(sp - 1 - (k)), SV(sp - 1 - (k)))

__device__ inline Val arith(u32 op, Val a, Val b) {
  if (a.t == CBH_T_DOUBLE && b.t == CBH_T_DOUBLE) {
    double x = as_f64(a.v), y = as_f64(b.v), r;
    switch (op) {
      case OP_ADD: r = x + y; break;
      case OP_SUB: r = x - y; break;
      case OP_MUL: r = x * y; break;
      case OP_DIV: r = x / y; break;
      default: return mk_errc(CBH_ERR_NO_SUCH_OVERLOAD);
    }
    return mk(CBH_T_DOUBLE, f64_bits(r));
  }
  if (a.t == CBH_T_INT && b.t == CBH_T_INT) {
    i64 x = (i64)a.v, y = (i64)b.v, r;
    switch (op) {
      case OP_ADD: if (__builtin_add_overflow(x, y, &r)) return mk_errc(CBH_ERR_INT_OVERFLOW); break;
      case OP_SUB: if (__builtin_sub_overflow(x, y, &r)) return mk_errc(CBH_ERR_INT_OVERFLOW); break;
      case OP_MUL: if (__builtin_mul_overflow(x, y, &r)) return mk_errc(CBH_ERR_INT_OVERFLOW); break;
      case OP_DIV: if (y == 0) return mk_errc(CBH_ERR_DIV_BY_ZERO); if (x == INT64_MIN && y == -1) return mk_errc(CBH_ERR_INT_OVERFLOW); r = x / y; break;
      case OP_MOD: if (y == 0) return mk_errc(CBH_ERR_MOD_BY_ZERO); if (x == INT64_MIN && y == -1) return mk_errc(CBH_ERR_INT_OVERFLOW); r = x % y; break;
      default: return mk_errc(CBH_ERR_NO_SUCH_OVERLOAD);
    }
    return mk(CBH_T_INT, (u64)r);
  }
  if (a.t == CBH_T_UINT && b.t == CBH_T_UINT) {
    u64 x = a.v, y = b.v, r;
    switch (op) {
      case OP_ADD: if (__builtin_add_overflow(x, y, &r)) return mk_errc(CBH_ERR_UINT_OVERFLOW); break;
      case OP_SUB: if (y > x) return mk_errc(CBH_ERR_UINT_OVERFLOW); r = x - y; break;
      case OP_MUL: if (__builtin_mul_overflow(x, y, &r)) return mk_errc(CBH_ERR_UINT_OVERFLOW); break;
      case OP_DIV: if (y == 0) return mk_errc(CBH_ERR_DIV_BY_ZERO); r = x / y; break;
      case OP_MOD: if (y == 0) return mk_errc(CBH_ERR_MOD_BY_ZERO); r = x % y; break;
      default: return mk_errc(CBH_ERR_NO_SUCH_OVERLOAD);
    }
    return mk(CBH_T_UINT, r);
  }
  // timestamps / durations
  i64 x = (i64)a.v, y = (i64)b.v, r;
  if (op == OP_ADD) {
    if ((a.t == CBH_T_TIMESTAMP && b.t == CBH_T_DURATION) || (a.t == CBH_T_DURATION && b.t == CBH_T_TIMESTAMP)) {
      if (__builtin_add_overflow(x, y, &r)) return mk_err();
      return mk(CBH_T_TIMESTAMP, (u64)r);
    }
    if (a.t == CBH_T_DURATION && b.t == CBH_T_DURATION) {
      if (__builtin_add_overflow(x, y, &r)) return mk_err();
      return mk(CBH_T_DURATION, (u64)r);
    }
  }
  if (op == OP_SUB) {
    if (a.t == CBH_T_TIMESTAMP && b.t == CBH_T_TIMESTAMP) { if (__builtin_sub_overflow(x, y, &r)) return mk_err(); return mk(CBH_T_DURATION, (u64)r); }
    if (a.t == CBH_T_TIMESTAMP && b.t == CBH_T_DURATION) { if (__builtin_sub_overflow(x, y, &r)) return mk_err(); return mk(CBH_T_TIMESTAMP, (u64)r); }
    if (a.t == CBH_T_DURATION && b.t == CBH_T_DURATION) { if (__builtin_sub_overflow(x, y, &r)) return mk_err(); return mk(CBH_T_DURATION, (u64)r); }
  }
  return mk_errc(CBH_ERR_NO_SUCH_OVERLOAD);  // (incl. string/list concatenation: not on the device)
}


// comparison / membership operators on two already-loaded values -> bool or error
__device__ inline Val compare_op(const Ctx& c, Lane& L, u32 op, Val x, Val y) {
  if (x.t == CBH_T_ERR) return x;
  if (y.t == CBH_T_ERR) return y;
  if (op == OP_EQ || op == OP_NE) {
    bool e = val_equal(c, L, x, y);
    return mk_bool(op == OP_EQ ? e : !e);
  }
  if (op == OP_IN) {
    bool found = false;
    if (y.t == CBH_T_LIST) {
      u32 n = cont_len(y.v);
      for (u32 i = 0; i < n && !found; ++i) found = val_equal(c, L, x, heap_get(c, cont_sel(y.v), cont_off(y.v) + i));
    } else if (y.t == CBH_T_MAP) {
      Val tmp; found = map_find(c, y, x, tmp);
    } else return mk_errc(CBH_ERR_NO_SUCH_OVERLOAD);
    return mk_bool(found);
  }
  int r = val_compare(c, x, y);
  if (r == 3) return mk_errc(CBH_ERR_NO_SUCH_OVERLOAD);
  bool res = false;
  if (r != 2) res = (op == OP_LT) ? r < 0 : (op == OP_LE) ? r <= 0 : (op == OP_GT) ? r > 0 : r >= 0;
  return mk_bool(res);
}

// Lane state crosses real calls BY VALUE (a reference would pin the caller's copy in scratch memory
// and turn every L.req / L.status access of the hot path into a memory round trip).
struct SlowVal { u32 t; u32 status; u64 v; };
// (inlining left to the compiler: forced out of line this costs the leaf kernels an AGPR and their fourth wave - hipcc's
// kernel-resource-usage remarks, tools/kernel_resources.py)
__device__ SlowVal compare_op_slow(const KernelArgs* ka, u32 req, u32 op, Val x, Val y) {
  VmLds none{};   // comparisons touch only the table / batch arrays
  const Ctx c = ctx_from_memory(ka, none);
  Lane L; L.req = req; L.edr = 0; L.status = 0; L.edr_err = false;
  const Val r = compare_op(c, L, op, x, y);
  return SlowVal{r.t, L.status, r.v};
}

// Everything the interpreter does WITH a rope (cbh_interp.h), out of line and by value like compare_op_slow.
//   kind 0 / 1  x == y / x != y          kind 2  x in y (list or map keys)      kind 3 / 4 / 5  startsWith / endsWith / contains
//   kind 6      size(x) in code points
enum { ROPE_EQ = 0, ROPE_NE = 1, ROPE_IN = 2, ROPE_STARTS = 3, ROPE_ENDS = 4, ROPE_CONTAINS = 5, ROPE_SIZE = 6 };
#ifndef CBH_HOSTSIM
__attribute__((noinline))
#endif
__device__ SlowVal rope_op(const KernelArgs* ka, const VmLds lds, u32 kind, Val x, Val y) {
  const Ctx c = ctx_from_memory(ka, lds);
  if (kind == ROPE_EQ || kind == ROPE_NE) {
    const bool e = sl_equal(c, x, y);
    return SlowVal{CBH_T_BOOL, 0, (u64)(e == (kind == ROPE_EQ))};
  }
  if (kind == ROPE_IN) {
    if (y.t != CBH_T_LIST && y.t != CBH_T_MAP) return SlowVal{CBH_T_ERR, 0, (u64)CBH_ERR_NO_SUCH_OVERLOAD};
    const u32 n = cont_len(y.v), step = y.t == CBH_T_MAP ? 2u : 1u;
    bool found = false;
    for (u32 i = 0; i < n && !found; ++i) found = sl_equal(c, x, heap_get(c, cont_sel(y.v), cont_off(y.v) + step * i));
    return SlowVal{CBH_T_BOOL, 0, (u64)found};
  }
  if (kind == ROPE_SIZE) {
    const u32 n = sl_len(c, x);
    u32 cp = 0;
    for (u32 i = 0; i < n; ++i) cp += (sl_byte(c, x, i) & 0xC0u) != 0x80u;
    return SlowVal{CBH_T_INT, 0, (u64)cp};
  }
  const u32 nh = sl_len(c, x), nn = sl_len(c, y);
  bool hit = false;
  if (nn <= nh) {
    u32 lo = 0, hi = nh - nn;
    if (kind == ROPE_STARTS) hi = 0;
    if (kind == ROPE_ENDS) lo = hi;
    for (u32 s = lo; s <= hi && !hit; ++s) {
      u32 j = 0;
      while (j < nn && sl_byte(c, x, s + j) == sl_byte(c, y, j)) ++j;
      hit = j == nn;
    }
  }
  return SlowVal{CBH_T_BOOL, 0, (u64)hit};
}

// Same-type fast paths of compare_op for the inline fused-leaf evaluation.
// Returns 1 / 0 = result, -1 = CEL error, -2 = not covered (caller takes the slow path).
__device__ __forceinline__ int fast_equal(Val x, Val y) {
  if (x.t == y.t) {
    if (x.t == CBH_T_DOUBLE) return as_f64(x.v) == as_f64(y.v);
    if (x.t < CBH_T_LIST || x.t == CBH_T_TIMESTAMP || x.t == CBH_T_DURATION) return x.v == y.v;
    return -2;   // containers
  }
  if (is_num(x.t) && is_num(y.t)) return -2;   // cross-type numeric equality
  return 0;                                     // different types are simply unequal
}
__device__ __forceinline__ int fast_compare(const Ctx& c, u32 op, Val x, Val y) {
  if (x.t == CBH_T_ERR || y.t == CBH_T_ERR) return -1;
  if (op == OP_EQ || op == OP_NE) {
    const int e = fast_equal(x, y);
    if (e < 0) return e;
    return (op == OP_EQ) ? e : 1 - e;
  }
  if (op == OP_IN) {
    if (y.t != CBH_T_LIST) return -2;
    const u32 n = cont_len(y.v), sel = cont_sel(y.v), off = cont_off(y.v);
    int found = 0;
    for (u32 i = 0; i < n; ++i) {
      const int e = fast_equal(x, heap_get(c, sel, off + i));
      if (e == -2) return -2;
      found |= e;
    }
    return found;
  }
  if (x.t == CBH_T_DOUBLE && y.t == CBH_T_DOUBLE) {
    const double a = as_f64(x.v), b = as_f64(y.v);
    return (op == OP_LT) ? a < b : (op == OP_LE) ? a <= b : (op == OP_GT) ? a > b : a >= b;   // NaN: all false
  }
  if (x.t == CBH_T_INT && y.t == CBH_T_INT) {
    const i64 a = (i64)x.v, b = (i64)y.v;
    return (op == OP_LT) ? a < b : (op == OP_LE) ? a <= b : (op == OP_GT) ? a > b : a >= b;
  }
  return -2;
}

// the tag of a cached column (`col` wave-uniform): the lane's byte of its dword in either form of the cache (CBH_CC_DWORDS)
__device__ __forceinline__ u32 cached_tag(const Ctx& c, u32 col, u32 req) {
  const bool packed = (c.flags & CBH_FI_PACKED_TAGS) != 0;
  const u32 tw = c.cc[(2u * c.n_cached + (packed ? col >> 2 : col)) * CBH_BLOCK + c.tid];
  const u32 byte = packed ? (col & 3u) : (u32)(((size_t)col * c.b.n_requests + req) & 3u);
  return (tw >> (byte * 8u)) & 0xFFu;
}
// a column of the kernel's LDS column cache (arg < n_cached)
__device__ __forceinline__ Val cached_column(const Ctx& c, const Lane& L, u32 arg) {
  const u32 t = cached_tag(c, arg, L.req);
  if (t == CBH_T_ABSENT) return mk_err();
  return mk(t, (u64)c.cc[arg * CBH_BLOCK + c.tid] | ((u64)c.cc[(c.n_cached + arg) * CBH_BLOCK + c.tid] << 32));
}

// operand of a fused leaf instruction (celc.py _simple_operand): 0 = constant, 1 = attribute column,
// 2 = request string field, 3 = attribute column known to sit in the column cache, 4 = principal id
__device__ __forceinline__ Val load_operand(const Ctx& c, const Lane& L, u32 kind, u32 arg) {
  if (kind == 3) return cached_column(c, L, arg);
  if (kind == 0) return mk(c.t.const_tag[arg], c.t.const_val[arg]);
  if (kind == 1) {
    if (arg < c.n_cached) return cached_column(c, L, arg);   // uniform: arg comes from the bytecode
    size_t ix = (size_t)arg * c.b.n_requests + L.req;
    u32 t = c.b.col_tag[ix];
    return t == CBH_T_ABSENT ? mk_err() : mk(t, c.b.col_val[ix]);
  }
  if (kind == 4) arg = CBH_RQ_PRINCIPAL_ID;
  return mk(CBH_T_STRING, c.b.req_u32[(size_t)arg * c.b.n_requests + L.req]);
}

