// cbh_walk2_kernel - the decision kernel for everything a table can hold: principal policies, role policies and parent
// roles, glob patterns, derived roles, conditions of any shape - in the flat kernel's form (cbh_check_flat.h): a wave
// walks scopes -> records ONCE, every lane carries all its role walks side by side as a bit vector (bit NA r + k = role
// r's walk for action k; three shapes NA x NR: eight actions and four roles in 32 bits, eight and eight or sixteen and four
// in 64; outside strict mode - anything else stays on cbh_check_wave.h), lanes merge by scope node deepest first,
// records are decided by class masks and glob bits.
//
// What differs from the flat kernel, which it generalises:
//   * Glob patterns.  A lane keeps, per action and per role, the match bits of its string for the first CBH_W2_MAX_GLOBS
//     patterns of the dimension (from the table's precomputed bits or cbh_resolve_globs_kernel's); a record with a glob in
//     a list carries the list's glob mask beside the class mask of its literals (CBH_SEC_ROWX).
//   * Parent roles.  A role slot is the class SET of [role] ++ ancestors (index.go:716-742), a 64-bit mask.
//   * Principal policies (check.go:195, the first pass): lanes differ in their principal, so the few lanes that have a
//     policy at all (a per-lane directory probe tells) are walked group by group before the resource walk; an action a
//     principal policy decides never enters the resource walk (check.go:445-448).
//   * Role policies (index.go:352-530): the reference walks a request role's [role] ++ ancestors list (ancestors sorted by
//     name).  At a scope with role policies every slot first meets the policy of its OWN role, then the roles that have a
//     policy there are visited in name order and a slot whose set holds one takes that rule's synthetic DENYs.  Index.Query's base test (index.go:250-305) comes from the union of the
//     bucket's role masks, kept in the directory.
//   * Conditions the walk cannot decide inline - generic programs, classified leaves that meet an int / uint / container
//     value - are EVALUATION SITES: cbh_walk2_pre_kernel, launched first when the table and batch have any, runs the same
//     walk without effects, evaluates every site a lane's roles and actions can reach with the shared evaluator
//     (cbh_check_wave.h eval_ref: the operand-stack interpreter and its ~230 registers live in THAT kernel) and leaves
//     four result bits per site and request; the walk reads them.  A site's slot is unique among what one request can
//     reach (lower/blob.py).  A site the reference would not have evaluated is evaluated in vain, never consulted.
//   * runtime.effectiveDerivedRoles reads the derived roles of the scope being walked (check.go:237-282): the pre-pass
//     evaluates a scope's definitions before that scope's sites.
// Same contract, same outputs, bit for bit, as cbh_check_wave.h.
#pragma once
#include "cbh_check_flat.h"

// Two waves to a workgroup (the flat kernels take four): measured on C5, 14.05 -> 14.5 G decisions/s (profiles/r06_ab_workgroup_waves.txt;
// eight waves lose 8 %).  The simulator's one-wave and the override builds follow the flat kernels' setting.
#if defined(CBH_HOSTSIM) || defined(CBH_FLAT_WAVES_OVERRIDE)
#define CBH_W2_WAVES CBH_FLAT_WAVES
#else
#define CBH_W2_WAVES 2u
#endif
#define CBH_W2_THREADS (CBH_W2_WAVES * CBH_BLOCK)
#define CBH_W2_LDS_GLOB_STRINGS 1024u   /* tables of at most this many strings keep their strings' glob match bits in LDS too */
#define CBH_W2_MAX_RP_ROLES 32u   /* roles with a role policy at one (version, scope) */

struct __attribute__((aligned(32))) TblRowX { u32 gslots, globs, rm_lo, rm_hi, am_lo, am_hi, p0, p1; };
struct __attribute__((aligned(64))) TblRpx { u32 resource, cnt, cond, gslot, am_lo, am_hi, ag, how; LeafRec leaf; };

// Dynamic LDS of one workgroup, per wave: [column cache][list arena (pre-pass of a table that builds lists)][scope chain]
// [per-action notes][site results (pre-pass)], then once per group the two class tables.
struct W2Layout { u32 cc_dw, arena_dw, chain_dw, aux_dw, gacc_dw, edr_dw, wave_dw, cls_bytes; };
#ifndef CBH_HOSTSIM
__host__ __device__
#endif
static inline W2Layout w2_layout(u32 ncc, bool arena, u32 table_max_depth, u32 table_scopes, bool pre, u32 n_gwords, u32 table_strings, u32 table_n_dr, u32 na = CBH_W2_NA, bool packed_tags = false) {
  W2Layout l;
  const u32 depth = table_max_depth < CBH_FLAT_MAX_DEPTH ? table_max_depth : CBH_FLAT_MAX_DEPTH;
  l.cc_dw = CBH_CC_DWORDS(ncc, packed_tags);
  l.arena_dw = (pre && arena) ? CBH_ARENA_ENTRIES * CBH_BLOCK * 9u / 4u : 0u;
  l.chain_dw = pre ? 0u : (table_scopes <= 256u ? depth * (CBH_BLOCK / 4u) : depth * CBH_BLOCK);
  l.aux_dw = pre ? 0u : na * CBH_BLOCK;
  l.gacc_dw = pre ? n_gwords * CBH_BLOCK * 2u : 0u;
  l.edr_dw = 0u; (void)table_n_dr;
  l.wave_dw = l.cc_dw + l.arena_dw + l.chain_dw + l.aux_dw + l.gacc_dw + l.edr_dw;
  // action class, role class, CBH_SWF_* per string; for a small table also the low 16 glob match bits of its strings (action, role)
  l.cls_bytes = table_strings <= CBH_FLAT_LDS_STRINGS ? (((3u * table_strings + 15u) & ~15u) + (table_strings <= CBH_W2_LDS_GLOB_STRINGS ? 4u * table_strings : 0u)) : 0u;
  return l;
}
static inline size_t w2_lds_bytes(const W2Layout& l, u32 waves) { return (size_t)l.wave_dw * 4u * waves + l.cls_bytes; }
// words of site results a launch needs: none for a table without sites, the generic sites only for a batch of plain scalars
static inline u32 w2_gwords(u32 gslots_generic, u32 gslots_all, bool plain_tags) {
  const u32 n = plain_tags ? gslots_generic : gslots_all;
  return (n + CBH_W2_SLOTS_PER_WORD - 1u) / CBH_W2_SLOTS_PER_WORD;
}

// The shape of a walk: NA actions x NR roles, one bit per (role, action) pair in a walk vector W - 32 bits for the base
// shape (8 x 4), 64 for the wider ones (8 x 8: the requests with five to eight roles; 16 x 4: those with nine to sixteen
// actions).  Bit NA r + k = role r's walk for action k.
template <bool WIDE> struct W2Word { typedef u32 type; };
template <> struct W2Word<true> { typedef u64 type; };
template <u32 NA, u32 NR> struct W2Shape {
  static_assert(NA * NR == 32u || NA * NR == 64u, "a walk vector is one or two dwords");
  static_assert(NA <= 16u && NR <= 8u, "glob bits of two actions to a dword; the role index of a note has three bits to spare");
  typedef typename W2Word<(NA * NR > 32u)>::type W;
  static constexpr W rep() { W m = 0; for (u32 r = 0; r < NR; ++r) m |= (W)1 << (NA * r); return m; }   // bit 0 of every role's field
};
__device__ __forceinline__ u32 w2_ctz(u32 x) { return (u32)__builtin_ctz(x); }
__device__ __forceinline__ u32 w2_ctz(u64 x) { return (u32)__builtin_ctzll(x); }
template <u32 NA, u32 NR>
__device__ __forceinline__ typename W2Shape<NA, NR>::W w2_rep_role(u32 rbits) {   // bit r -> all NA bits of field r
  typedef typename W2Shape<NA, NR>::W W;
  if (NA == 8u && NR == 4u) {
    const u32 x = rbits & 0xFu;
    return (W)(((x | (x << 7) | (x << 14) | (x << 21)) & 0x01010101u) * 0xFFu);
  }
  W m = 0;
#pragma unroll
  for (u32 r = 0; r < NR; ++r) m |= (W)((rbits >> r) & 1u) << (NA * r);
  return m * (W)((1u << NA) - 1u);
}

// (one reservation of n consecutive entries of a site's list, by one lane)
__device__ __forceinline__ u32 w2_list_reserve(CBH_G u32* cnt, u32 n) {
#ifdef CBH_HOSTSIM
  const u32 old = *cnt; *cnt += n; return old;
#else
  return atomicAdd((unsigned int*)cnt, n);
#endif
}
// PMODE: 0 = the walk; 1 = the pre-pass (the sites a request reaches are evaluated where the walk meets them); 2 = the pre-pass's
// COLLECTOR (cbh_walk2_collect_kernel: the same walk, no evaluator - every (request, site) met is appended to the site's list for
// cbh_walk2_interp_kernel to evaluate on full waves)
// EP (the walk only, cbh_walk2_trail_kernel*): AuditTrail.EffectivePolicies of the batch (cbh_check_batch_trail).  The reference meets a
// request's roles one after the other and stops behind the first that allows (check.go:208-442); this walk takes them side by side.
// So the walk runs TWICE: once as always - that tells, per action, the first allowing role -, then again for the walks the reference
// really makes (the roles up to that one), and every binding that second walk iterates marks its policy (check.go:302-304).  The fold
// reads the second walk's results: for the walks it is restricted to they are the first's.
template <int PMODE, u32 NA_, u32 NR_, bool EP = false>
__device__ __forceinline__ void w2_body(const KernelArgs& ka_regs, Ctx& c, const W2Layout& ly) {
  constexpr bool PRE = PMODE != 0;
  static_assert(!(EP && PRE), "the trail is the walk's");
  const TableDev& t = ka_regs.t;
  const BatchDev& b = ka_regs.b;
  const OutDev& o = ka_regs.o;
  const u32 flags = ka_regs.flags;
  const u32 wave = threadIdx.x / CBH_BLOCK;   // (NOT through readfirstlane: measured 7-19 % slower on every workload, profiles/r06_ab_prologue.txt)
  constexpr u32 NA = NA_, NR = NR_;
  typedef typename W2Shape<NA, NR>::W W;
  constexpr W REP = W2Shape<NA, NR>::rep();       // x * REP: an action mask in every role's field
  constexpr u32 AMASK = (1u << NA) - 1u;
#ifdef CBH_PROFILE_CYCLES   // profiling build only (tools/gpu_cycles_walk2.py): per-wave phase cycles into the policy / scope words
  const u64 cyc0 = __builtin_readcyclecounter();
  const u64 rt0 = __builtin_amdgcn_s_memrealtime();
  u64 cyc_eval = 0; u32 dbg_rows = 0, dbg_rounds = 0, dbg_evals = 0;
#define W2_DBG(x) x
#else
#define W2_DBG(x)
#endif
  const u32 rix = b.req_lo + blockIdx.x * (PRE ? CBH_BLOCK : CBH_W2_THREADS) + threadIdx.x;
  const u32 NRQ = b.n_requests;
  bool valid = rix < b.req_hi;
  if ((flags & CBH_FI_SKIP_WIDE) && valid) {   // wave-uniform test first: the two loads only for a batch that has wider requests
    const u32 na = b.req_u32[(size_t)CBH_RQ_ACT_CNT * NRQ + rix], nr = b.req_u32[(size_t)CBH_RQ_ROLE_CNT * NRQ + rix];
    valid = cbh_w2_class(na, nr) == (NA > CBH_W2_NA ? 2u : NR > CBH_W2_NR ? 1u : 0u);   // the requests of this shape's class
  }
  const u32 req = valid ? rix : b.req_lo;
  const bool has_pp = (t.flags & CBH_MF_HAS_PRINCIPAL_POLICIES) != 0;
  const bool has_parents = (t.flags & CBH_MF_HAS_PARENT_ROLES) != 0;
  const bool has_rolepol = (t.flags & CBH_MF_HAS_ROLE_POLICIES) != 0;
  const bool aglobs = t.nfa_words[DIM_ACTION] != 0, rglobs = t.nfa_words[DIM_ROLE] != 0;
#define RQ(f) b.req_u32[(size_t)(f) * NRQ + req]
  const u32 pid = RQ(CBH_RQ_PRINCIPAL_ID), kind = RQ(CBH_RQ_KIND), r_scope = RQ(CBH_RQ_R_SCOPE), r_ver = RQ(CBH_RQ_R_VERSION);
  const u32 role_off = RQ(CBH_RQ_ROLE_OFF), act_off = RQ(CBH_RQ_ACT_OFF);
  const u32 role_cnt = valid ? RQ(CBH_RQ_ROLE_CNT) : 0, act_cnt = valid ? RQ(CBH_RQ_ACT_CNT) : 0;   // <= NR / <= NA (host-checked, or filtered above)
  // (unconditional: a load inside a conditional block is waited for at the block's end - with everything issued before it)
  const u32 p_scope = RQ(CBH_RQ_P_SCOPE), p_ver = RQ(CBH_RQ_P_VERSION);
#undef RQ
  // The request's loads in two round trips (cbh_check_flat.h flat_body, cbh_check_wave.h cc_load_tags / cc_fill): first what
  // depends on nothing - the request words above, the speculated action ids, the class tables' bytes, the columns' tag bytes -,
  // then the role ids and, LAST, behind this prologue's own LDS stores, the columns' asynchronous copies.
  const u32 w0r = b.req_lo + blockIdx.x * (PRE ? CBH_BLOCK : CBH_W2_THREADS) + wave * CBH_BLOCK;
  const u32 w0 = w0r < b.req_hi ? w0r : b.req_lo;   // the wave's first request (uniform)
  const u32 wd = valid ? c.tid : 0u;
  const bool spec = b.n_tuples >= 4u;
  const u32 spec_ix = (4u * req + 4u <= b.n_tuples) ? 4u * req : 0u;
  const u32x4u sp = load_u32x4(spec ? b.tuple_action + spec_ix : b.req_u32);
  // this thread's entries of the tables the workgroup keeps in LDS (classes, string flags, glob bits): first-trip loads too - IN FRONT of
  // the tag bytes, whose block (packed form only) ends in a full wait
  const u32 kmax = t.K ? t.K - 1u : 0u;
  const u32 tbl_i = threadIdx.x < t.K ? threadIdx.x : kmax;
  const CBH_G u8* safe8 = (const CBH_G u8*)b.req_u32;   // (a table without strings / without glob bits: the unconditional loads read a request word)
  const u32 tbl_a0 = (t.K ? t.action_class : safe8)[tbl_i], tbl_r0 = (t.K ? t.role_class : safe8)[tbl_i], tbl_w0 = (t.K ? t.str_wflags : safe8)[tbl_i];
  const CBH_G u64* gb_base = (t.K && t.gbits && (aglobs || rglobs)) ? t.gbits : (const CBH_G u64*)b.req_u32;
  const bool gb_real = gb_base == t.gbits && t.gbits != nullptr;
  const u64 tbl_ga0 = gb_base[gb_real ? (size_t)DIM_ACTION * t.K + tbl_i : 0u], tbl_gr0 = gb_base[gb_real ? (size_t)DIM_ROLE * t.K + tbl_i : 0u];
  const CcTags cct = cc_load_tags(c, b, NRQ, w0, wd);
  const bool lenient = (flags & CBH_F_LENIENT_SCOPE_SEARCH) != 0;
  // the chain's first scope (per lane, reads the scope tables; needs the request words): the pre-pass forms ask about it at once,
  // the walk issues its load with the second trip, behind the columns' copies
  u32 first = CBH_NONE;
  if (PRE) first = chain_first(t, r_scope, FLAG_RES, lenient);
  bool pre_climbs = false;   // pre-pass: does anything on this request's path hold a site the batch files?
  bool pre_family = false, pre_other = false, pre_pp = false;   // ... its family's records / definitions; role-policy rules; principal policies
  u64 pre_fam_roles = ~0ull;   // the role classes that reach a generic site of the family (CBH_B_FAMILY v1, v2)
  if (PRE) {
    const u32 filed = (b.n_gslots ? (CBH_BS_ROW_GENERIC | CBH_BS_DR_GENERIC) : 0u) | (b.n_gslots > t.gslots_generic ? (CBH_BS_ROW_OPEN | CBH_BS_DR_OPEN) : 0u);
    uint4 fv; fv.x = 0; fv.y = 0; fv.z = 0; fv.w = 0;
    const bool has_walks = valid && role_cnt != 0 && act_cnt != 0;
    if (has_walks && first != CBH_NONE && dir_find(t, CBH_B_FAMILY, r_ver, kind, 0, fv)) pre_family = (fv.x & filed) != 0;
    if (pre_family && (fv.w & 1u) && b.n_gslots <= t.gslots_generic) pre_fam_roles = (u64)fv.y | ((u64)fv.z << 32);   // (only generic sites filed: the mask is theirs)
    pre_other = has_walks && (t.q_sites & filed & (CBH_BS_ROW_GENERIC | CBH_BS_ROW_OPEN)) != 0;   // role-policy rules: any request of the version may reach them
    pre_climbs = pre_family || pre_other;
    pre_pp = has_walks && has_pp && (t.q_sites & filed & (CBH_BS_DR_GENERIC | CBH_BS_DR_OPEN)) != 0;   // principal policies with sites
    if (wave_ballot(pre_climbs || pre_pp) == 0) return;   // nothing to evaluate for this wave
  }
  W2_DBG(const u64 cycA0 = __builtin_readcyclecounter();)
  W2_DBG(const u64 cycA = __builtin_readcyclecounter();)
  const u32 all = (1u << act_cnt) - 1u;
  const u32 max_depth = t.max_depth < CBH_FLAT_MAX_DEPTH ? t.max_depth : CBH_FLAT_MAX_DEPTH;
  const bool chain8 = t.n_scopes <= 256u;
  CBH_L u32* wave_lds = (CBH_L u32*)cbh_dyn_lds + wave * ly.wave_dw;
  CBH_L u32* chain_si = wave_lds + ly.cc_dw + ly.arena_dw;
  CBH_L u8* chain_si8 = (CBH_L u8*)chain_si;
  CBH_L u32* aux = chain_si + ly.chain_dw;                                   // [action][lane]: see the fold
  CBH_L u64* gacc = (CBH_L u64*)(wave_lds + ly.cc_dw + ly.arena_dw + ly.chain_dw + ly.aux_dw);   // [word][lane] (pre-pass)
  CBH_L u8* cls_lds = (CBH_L u8*)((CBH_L u32*)cbh_dyn_lds + (PRE ? 1u : CBH_W2_WAVES) * ly.wave_dw);
  const bool cls_in_lds = ly.cls_bytes != 0;
  const bool gb_in_lds = cls_in_lds && t.K <= CBH_W2_LDS_GLOB_STRINGS && (aglobs || rglobs);
  CBH_L unsigned short* gb_lds = (CBH_L unsigned short*)(cls_lds + ((3u * t.K + 15u) & ~15u));   // [action bits K][role bits K]
  if (cls_in_lds) {
    if (threadIdx.x < t.K) {
      cls_lds[threadIdx.x] = (u8)tbl_a0; cls_lds[t.K + threadIdx.x] = (u8)tbl_r0; cls_lds[2u * t.K + threadIdx.x] = (u8)tbl_w0;
      if (gb_in_lds) {
        gb_lds[threadIdx.x] = aglobs ? (unsigned short)tbl_ga0 : (unsigned short)0;
        gb_lds[t.K + threadIdx.x] = rglobs ? (unsigned short)tbl_gr0 : (unsigned short)0;
      }
    }
    for (u32 i = threadIdx.x + (PRE ? CBH_BLOCK : CBH_W2_THREADS); i < t.K; i += (PRE ? CBH_BLOCK : CBH_W2_THREADS)) {
      cls_lds[i] = t.action_class[i]; cls_lds[t.K + i] = t.role_class[i]; cls_lds[2u * t.K + i] = t.str_wflags[i];
      if (gb_in_lds) {
        gb_lds[i] = aglobs ? (unsigned short)t.gbits[(size_t)DIM_ACTION * t.K + i] : (unsigned short)0;
        gb_lds[t.K + i] = rglobs ? (unsigned short)t.gbits[(size_t)DIM_ROLE * t.K + i] : (unsigned short)0;
      }
    }
  }
  if (PRE) { for (u32 w = 0; w < b.n_gwords; ++w) gacc[w * CBH_BLOCK + c.tid] = 0; }
  else {
#pragma unroll
    for (u32 k = 0; k < NA; ++k) aux[k * CBH_BLOCK + c.tid] = CBH_NONE;
  }

  // ---- actions and roles: ids -> classes (63 = a string no rule names) and glob match bits
  u32 aid[NA], rid[NR], ac[NA];
#pragma unroll
  for (u32 k = 0; k < NR; ++k) rid[k] = b.roles[k < role_cnt ? role_off + k : 0u];
  const bool spec_hit = spec && act_cnt == 4u && act_off == spec_ix;
  aid[0] = sp.x; aid[1] = sp.y; aid[2] = sp.z; aid[3] = sp.w;
  if (!spec_hit) {
#pragma unroll
    for (u32 k = 0; k < 4; ++k) aid[k] = b.tuple_action[k < act_cnt ? act_off + k : 0u];
  }
#pragma unroll
  for (u32 k = 4; k < NA; ++k) aid[k] = 0;
  if (wave_ballot(act_cnt > 4u) != 0) {
#pragma unroll
    for (u32 k = 4; k < NA; ++k) aid[k] = b.tuple_action[k < act_cnt ? act_off + k : 0u];
  }
  cc_fill(c, b, NRQ, w0, wd, cct);   // (behind every LDS store of this prologue)
  if (!PRE) first = chain_first(t, r_scope, FLAG_RES, lenient);
  u32 rcls[NR];
  u32 rpar = 0;   // bit r: the role has ancestors in some scope (CBH_SWF_PARENTS): the only ones the directory is asked about
  if (cls_in_lds) __syncthreads();
#pragma unroll
  for (u32 k = 0; k < NA; ++k) {
    const u32 ix = aid[k] < t.K ? aid[k] : kmax;
    const u32 ca = cls_in_lds ? (u32)cls_lds[ix] : (u32)t.action_class[ix];
    ac[k] = (k < act_cnt && aid[k] < t.K && ca < 62u) ? ca : 63u;
  }
#pragma unroll
  for (u32 k = 0; k < NR; ++k) {
    const u32 ix = rid[k] < t.K ? rid[k] : kmax;
    const u32 cr = cls_in_lds ? (u32)cls_lds[t.K + ix] : (u32)t.role_class[ix];
    rcls[k] = (k < role_cnt && rid[k] < t.K && cr < 62u) ? cr : 63u;
    if (has_parents && k < role_cnt && rid[k] < t.K) rpar |= (((cls_in_lds ? (u32)cls_lds[2u * t.K + ix] : (u32)t.str_wflags[ix]) & CBH_SWF_PARENTS) ? 1u : 0u) << k;
  }
  // does the principal have a principal policy at all (some version, some scope)?  Else the first pass has nothing to walk
  const bool pid_has_pp = has_pp && pid < t.K && (((cls_in_lds ? (u32)cls_lds[2u * t.K + pid] : (u32)t.str_wflags[pid]) & CBH_SWF_PRINCIPAL) != 0);
  u32 gap[NA / 2], rgp[NR / 2];   // glob match bits, two 16-bit fields to a dword
#pragma unroll
  for (u32 j = 0; j < NA / 2; ++j) gap[j] = 0;
#pragma unroll
  for (u32 j = 0; j < NR / 2; ++j) rgp[j] = 0;
  if (aglobs) {
#pragma unroll
    for (u32 k = 0; k < NA; ++k) {
      const u32 g = k >= act_cnt ? 0u : (gb_in_lds && aid[k] < t.K) ? (u32)gb_lds[aid[k]] : ((u32)gbits_of(t, b, DIM_ACTION, aid[k]) & 0xFFFFu);
      gap[k >> 1] |= g << (16u * (k & 1u));
    }
  }
  if (rglobs) {
#pragma unroll
    for (u32 k = 0; k < NR; ++k) {
      const u32 g = k >= role_cnt ? 0u : (gb_in_lds && rid[k] < t.K) ? (u32)gb_lds[t.K + rid[k]] : ((u32)gbits_of(t, b, DIM_ROLE, rid[k]) & 0xFFFFu);
      rgp[k >> 1] |= g << (16u * (k & 1u));
    }
  }
  W2_DBG(const u64 cycB = __builtin_readcyclecounter();)
  // role slots as class sets: [role] ++ ancestors for the request's own resource scope (check.go:172, 227)
  u32 rs_lo[NR], rs_hi[NR];
#pragma unroll
  for (u32 k = 0; k < NR; ++k) {
    const u64 m = k < role_cnt ? (1ull << rcls[k]) : 0ull;
    rs_lo[k] = (u32)m; rs_hi[k] = (u32)(m >> 32);
  }
  if (has_parents) {
    const u32 pr_scope_key = (r_scope & CBH_SCOPE_EXACT) ? (r_scope & ~CBH_SCOPE_EXACT) : CBH_NONE;
#pragma unroll
    for (u32 k = 0; k < NR; ++k) {
      uint4 pv;
      if (((rpar >> k) & 1u) && pr_scope_key != CBH_NONE && dir_find(t, CBH_B_PARENTS, pr_scope_key, rid[k], 0, pv)) {
        rs_lo[k] |= pv.z; rs_hi[k] |= pv.w;   // the OR of the ancestors' classes travels in the entry
        if (rglobs) for (u32 j = 0; j < pv.y; ++j)   // (their role-glob bits: ancestors are table strings)
          rgp[k >> 1] |= ((u32)t.gbits[(size_t)DIM_ROLE * t.K + t.pool[pv.x + j]] & 0xFFFFu) << (16u * (k & 1u));
      }
    }
  }
  W2_DBG(const u64 cycC = __builtin_readcyclecounter();)
  u32 lane_rs_lo = 0, lane_rs_hi = 0, lane_ac_lo = 0, lane_ac_hi = 0;
  W walks = 0;   // bit NA r + k: role r exists and action k exists
#pragma unroll
  for (u32 k = 0; k < NA; ++k) {
    if (k < act_cnt) { const u64 m = 1ull << ac[k]; lane_ac_lo |= (u32)m; lane_ac_hi |= (u32)(m >> 32); }
  }
#pragma unroll
  for (u32 k = 0; k < NR; ++k) {
    if (k < role_cnt) { lane_rs_lo |= rs_lo[k]; lane_rs_hi |= rs_hi[k]; walks |= (W)all << (NA * k); }
  }
  if (PRE) {
    // the role sets are known now ([role] ++ ancestors, as classes): a request none of whose roles reaches a generic site of its
    // family has nothing to evaluate there - and as requests are grouped by route and role list, whole waves leave here
    if (pre_family && !rglobs && (((u64)lane_rs_lo | ((u64)lane_rs_hi << 32)) & pre_fam_roles) == 0) pre_family = false;
    pre_climbs = pre_family || pre_other;
    if (wave_ballot(pre_climbs || pre_pp) == 0) return;
  }
  // classes / glob bits present in the wave: a record none of them can match is skipped on the scalar unit
  const u64 wave_a = wave_or64((u64)lane_ac_lo | ((u64)lane_ac_hi << 32), wave, c.tid);
  const u64 wave_r = wave_or64((u64)lane_rs_lo | ((u64)lane_rs_hi << 32), wave, c.tid);
  const u32 wave_ac_lo = (u32)wave_a, wave_ac_hi = (u32)(wave_a >> 32), wave_rc_lo = (u32)wave_r, wave_rc_hi = (u32)(wave_r >> 32);

  u64 edr_scope = 0;   // pre-pass: the derived roles of the scope being walked (what runtime.effectiveDerivedRoles reads)

  // A condition reference for the lanes with `active`: bit 0 satisfied, bit 1 CEL error, bit 3 outside the device subset.
  // `how`: 1 = `lr` is the fused leaf, 2 = `lr` describes a tree of classified leaves (both inline, cbh_check_flat.h), 0 =
  // neither.  What the inline code leaves open is an evaluation site, slot `gslot`: the pre-pass evaluates it with the
  // shared evaluator and files the result, the walk reads it.
  auto leafish = [&](u32 ref, u32 how, const LeafRec& lr, u32 gslot, bool active) -> u32 {
    u32 lv = 4u;
    if (how == 1u) lv = flat_leaf(c, lr, req, pid);
    else if (how == 2u) lv = flat_tree(c, lr, req, pid);
    const bool slow = active && lv == 4u;
    if (wave_ballot(slow) != 0) {
      const bool filed = gslot < b.n_gslots;   // uniform (CBH_GSLOT_NONE is beyond any count)
      if (PMODE == 2) {
        // the collector: the site goes on its slot's list once per request (bit 2 of the slot's result nibble says it did; the
        // walk reads the nibble through 0xB), its outcome is taken as "satisfied" so that what lies behind it is collected too
        const u32 nib = 4u * (gslot % CBH_W2_SLOTS_PER_WORD);
        const bool add = slow && filed && !((gacc[(gslot / CBH_W2_SLOTS_PER_WORD) * CBH_BLOCK + c.tid] >> (nib + 2u)) & 1ull);
        const u64 who = wave_ballot(add);
        if (who != 0) {
          u32 base = 0;
          if (c.tid == first_lane(who)) base = w2_list_reserve(b.site_cnt + gslot, (u32)__builtin_popcountll(who));
          base = wave_readlane(base, first_lane(who));
          if (add) {
            const u32 rank = (u32)__builtin_popcountll(who & ((1ull << c.tid) - 1ull));
            b.site_list[(size_t)gslot * b.site_cap + base + rank] = (u64)req | ((u64)ref << 32);
            gacc[(gslot / CBH_W2_SLOTS_PER_WORD) * CBH_BLOCK + c.tid] |= 4ull << nib;
          }
        }
        if (slow) lv = filed ? 1u : 8u;
      } else if (PRE) {
        W2_DBG(const u64 e0 = __builtin_readcyclecounter(); ++dbg_evals;)
        const u32 r = eval_ref<true>(c.ka_mem, lds_of(c), req, edr_scope, false, ref, slow);
        W2_DBG(cyc_eval += __builtin_readcyclecounter() - e0;)
        const u32 st = r >> 8;
        const u32 code = ((r & 0xFFu) == 1u ? 1u : 0u) | ((st & CBH_ST_CEL_ERROR) ? 2u : 0u) | ((st & CBH_ST_UNSUPPORTED) ? 8u : 0u);
        if (slow) {
          lv = code;
          if (filed) gacc[(gslot / CBH_W2_SLOTS_PER_WORD) * CBH_BLOCK + c.tid] |= (u64)code << (4u * (gslot % CBH_W2_SLOTS_PER_WORD));
        }
      } else if (slow) {
        // (a site without a filed result - the host launches the pre-pass whenever one can be needed - is loud, never guessed)
        lv = filed ? (u32)(b.gres[(size_t)(gslot / CBH_W2_SLOTS_PER_WORD) * NRQ + req] >> (4u * (gslot % CBH_W2_SLOTS_PER_WORD))) & 0xBu : 8u;
      }
    }
    return active ? lv : 0u;
  };
  auto act_bits16 = [&](u32 k) -> u32 { return (gap[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu; };
  // the request's actions (bit k) a class mask + glob mask admits
  auto match_actions = [&](u32 am_lo, u32 am_hi, u32 ag) -> u32 {
    const u64 am = (u64)am_lo | ((u64)am_hi << 32);
    u32 m = 0;
#pragma unroll
    for (u32 k = 0; k < NA; ++k) m |= (u32)((am >> ac[k]) & 1ull) << k;
    if (ag) {
#pragma unroll
      for (u32 k = 0; k < NA; ++k) m |= ((act_bits16(k) & ag) != 0u ? 1u : 0u) << k;
    }
    return m & all;
  };
  auto match_roles = [&](u32 rm_lo, u32 rm_hi, u32 rg) -> u32 {   // bit r
    u32 m = 0;
#pragma unroll
    for (u32 k = 0; k < NR; ++k) m |= (((rm_lo & rs_lo[k]) | (rm_hi & rs_hi[k])) != 0u ? 1u : 0u) << k;
    if (rg) {
#pragma unroll
      for (u32 k = 0; k < NR; ++k) m |= ((((rgp[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu) & rg) != 0u ? 1u : 0u) << k;
    }
    return m;
  };
  auto kind_bits = [&]() -> u64 { return gbits_of(t, b, DIM_KIND, kind); };

  W err = 0, unsup = 0;   // walk bits whose evaluation met a CEL error / left the device subset
  W wtr = 0;                // walk bits that visited a rule with output expressions, or whose variables the device could not evaluate:
                            // the trace pass has (or may have) something to say about the input (CBH_ST_WANTS_TRACE)
  const LeafRec no_leaf{};
  // the probe of a params set's variables (celc.py vars_probe_program; slot CBH_GSLOT_NONE = the set has none): bit 1 = one of
  // them failed - an evaluation error whether or not anything reads it (check.go:651-677) -, bit 3 = the device cannot tell
  auto probe = [&](u32 slot, u32 pc_off, u32 which, bool active) -> u32 {
    if (slot == CBH_GSLOT_NONE) return 0u;   // uniform
    const u32 pc = (PRE && pc_off != CBH_NONE) ? uload(&t.pool[pc_off + which]) : 0u;
    return leafish(pc, 0u, no_leaf, slot, active);
  };
  W2_DBG(const u64 cyc1 = __builtin_readcyclecounter();)   // request fields, ids, classes, role sets are there

  const bool want_ep = EP && (flags & CBH_F_WANT_EFFECTIVE_POLICIES) != 0 && o.eff_pol != nullptr;
  const u32 n_pass = want_ep ? 2u : 1u;
  const W walks_all = walks;
  bool marks = false;   // this lane's visits mark their policies (the second walk of a request that has anything to evaluate)
  u32 marked = CBH_NONE;   // ... the policy it marked last: a bucket's rules are one policy's, the mark is made once per bucket
  auto mark = [&](u32 policy) { if (policy != marked) { ep_mark(o, b, req, policy); marked = policy; } };
  u32 p_allow = 0, p_deny = 0, p_err = 0, p_unsup = 0, p_wtr = 0, p_pol = 0;
  u32 p_first = CBH_NONE;
  u32 p_done = 0;
  W S = 0, has_allow = 0, allow = 0, deny = 0;
  W dp0 = 0, dp1 = 0, dp2 = 0, dp3 = 0;
  const u32 scope_bits = t.n_scopes > 1 ? 32u - (u32)__builtin_clz(t.n_scopes - 1u) : 0u;
  u32 cur = first, mydepth = 0;
  bool exists = false;
  const bool pre_edr = PRE && (t.flags & CBH_MF_USES_RUNTIME_EDR) != 0;
  const bool want_edr = !PRE && (flags & CBH_F_WANT_DERIVED_ROLES) != 0 && t.n_dr != 0;
  u64 edr_all = 0; bool derr_all = false, dunsup_all = false, dwtr_all = false;   // derived roles over the chain positions visited with a walk still going
  u32 edr_vis = 0;                                              // ... how many those were
  // nothing to evaluate at all?  (check.go:116-121, 165-170; after a walk: `exists` is its finding)
  bool p_exists = false;
  auto nothing_to_evaluate = [&]() -> bool {
    p_exists = false;
    if (has_pp && valid && !exists && p_first != CBH_NONE) {
      for (u32 si = p_first; si != CBH_NONE && !p_exists; si = chain_next(t, t.scope_parent[si], FLAG_PRIN)) {
        uint4 v;
        p_exists = dir_find(t, CBH_B_PPEXISTS, p_ver, si, 0, v);
      }
    }
    return (p_first == CBH_NONE && first == CBH_NONE) || (!p_exists && !exists);
  };
  W2_DBG(u64 cycP = cyc1; u64 cyc2 = cyc1;)
  for (u32 pass = 0; pass < n_pass; ++pass) {
  if (EP && pass == 1u) {
    // the walks the reference really makes: per action the roles up to the first that allowed (all of them when none did)
    W legit = 0;
#pragma unroll
    for (u32 k = 0; k < NA; ++k) {
      const W ak = (allow >> k) & REP;
      const W seen = ak ? (((ak & ((W)0 - ak)) << 1) - 1u) : ~(W)0;
      legit |= ((walks_all >> k) & REP & seen) << k;
    }
    marks = valid && !nothing_to_evaluate();
    walks = walks_all & legit;
    p_allow = p_deny = p_err = p_unsup = p_wtr = p_pol = 0; p_first = CBH_NONE;
    has_allow = allow = deny = 0; dp0 = dp1 = dp2 = dp3 = 0;
    cur = first; mydepth = 0; exists = false;
    edr_all = 0; derr_all = dunsup_all = dwtr_all = false; edr_vis = 0;
    err = unsup = wtr = 0;
#pragma unroll
    for (u32 k = 0; k < NA; ++k) aux[k * CBH_BLOCK + c.tid] = CBH_NONE;
  }
  if (has_pp && (!PRE || (t.q_sites & (CBH_BS_DR_GENERIC | CBH_BS_DR_OPEN)) != 0)) {   // (pre-pass: only when principal policies hold sites at all)
    p_first = chain_first(t, p_scope, FLAG_PRIN, lenient);
    const bool cand = valid && pid_has_pp && p_first != CBH_NONE && role_cnt > 0 && act_cnt > 0;
    bool pend = false;
    if (cand) {   // per lane, all lanes at once: is there a policy of this principal on the chain, and does a principal policy "exist" (check.go:216-225)
      bool pe = false;
      for (u32 si = p_first; si != CBH_NONE && !(pend && pe); si = chain_next(t, t.scope_parent[si], FLAG_PRIN)) {
        uint4 v;
        pend = pend || dir_find(t, CBH_B_PRINCIPAL, r_ver, si, pid, v);   // the resource's version: check.go:294
        pe = pe || dir_find(t, CBH_B_PPEXISTS, p_ver, si, 0, v);
      }
      p_pol = pe ? (((u32)CBH_P_PRINCIPAL << 28) | p_first) : ((u32)CBH_P_NO_MATCH << 28);
    }
    W2_DBG(cycP = __builtin_readcyclecounter();)
    for (;;) {
      const u64 rem = wave_ballot(pend);
      if (rem == 0) break;
      const u32 lead = first_lane(rem);
      const u32 g_first = wave_readlane(p_first, lead), g_ver = wave_readlane(r_ver, lead), g_pid = wave_readlane(pid, lead);
      const bool ing = pend && p_first == g_first && r_ver == g_ver && pid == g_pid;
      pend = pend && !ing;
      u32 S = ing ? all : 0u, has_allow = 0;
      for (u32 si = g_first; si != CBH_NONE; si = uchain_next(t, uload(&t.scope_parent[si]), FLAG_PRIN)) {
        if (wave_ballot(S != 0) == 0) break;
        uint4 bucket; bucket.x = bucket.y = 0;
        if (udir_find(t, CBH_B_PRINCIPAL, g_ver, si, g_pid, bucket)) {
          for (u32 row = bucket.x; row < bucket.x + bucket.y; ++row) {
            const TblRowFull rf = uload_rec<TblRowFull>(t.rows, row);
            const TblRow& rw = rf.hot;
            const TblRowPat pt = uload_rec<TblRowPat>(t.rowpat, row);
            const u64 kb = t.nfa_words[DIM_KIND] ? kind_bits() : 0ull;
            u32 mrow = 0;
            if (S != 0 && pat_match(pt.resource, kind, kb)) {
              const u32 n_act = pt.counts & 0xFFFFu;   // 0 = a single inline reference
              auto one = [&](u32 pat) -> u32 {
                if (pat == CBH_PAT_ANY) return all;
                u32 m = 0;
                if (pat & CBH_PAT_GLOB) {
                  const u32 gi = pat & 15u;
#pragma unroll
                  for (u32 k = 0; k < NA; ++k) m |= ((act_bits16(k) >> gi) & 1u) << k;
                } else {
#pragma unroll
                  for (u32 k = 0; k < NA; ++k) m |= (u32)(aid[k] == pat) << k;
                }
                return m & all;
              };
              if (rw.flags & CBH_ROW_F_ACTION_LIST) { for (u32 i = 0; i < n_act; ++i) mrow |= one(uload(&t.pool[pt.action + i])); }
              else { mrow = one(pt.action); if (n_act > 1) mrow |= one(pt.a1); if (n_act > 2) mrow |= one(pt.a2); }
              mrow &= PRE ? all : S;
            }
            if (wave_ballot(mrow != 0) == 0) continue;
            if (EP && marks && mrow != 0) mark(rw.policy);   // the binding is iterated (check.go:302-304)
            u32 hit = mrow;
            u32 gslot = CBH_GSLOT_NONE;
            if (rw.flags & CBH_ROW_F_OUTPUT) p_wtr |= mrow;
            if (rw.flags & CBH_ROW_F_X) {
              const TblRowX rx = uload_rec<TblRowX>(t.rowx, row);
              gslot = rx.gslots & 0xFFFFu;
              const u32 pv = probe(rx.p0 & 0xFFFFu, rx.p1, 0u, mrow != 0);
              p_err |= (pv & 2u) ? mrow : 0u; p_wtr |= (pv & 8u) ? mrow : 0u;
            }
            if (rw.cond != CBH_NONE) {
              const u32 how = (rw.flags & CBH_ROW_F_LEAF_EMBEDDED) ? 1u : (rw.flags & CBH_ROW_F_TREE_EMBEDDED) ? 2u : 0u;
              const u32 lv = leafish(rw.cond, how, rf.leaf, gslot, hit != 0);
              p_err |= (lv & 2u) ? hit : 0u; p_unsup |= (lv & 8u) ? hit : 0u;
              hit = (lv & 1u) ? hit : 0u;
            }
            if (PRE) continue;
            if ((rw.flags & 3u) == CBH_EFFECT_ALLOW) has_allow |= hit;
            else if ((rw.flags & 3u) == CBH_EFFECT_DENY && hit) {
              p_deny |= hit; S &= ~hit;
#pragma unroll
              for (u32 k = 0; k < NA; ++k) if ((hit >> k) & 1u) aux[k * CBH_BLOCK + c.tid] = si;
            }
          }
        }
        if (PRE) continue;
        const u32 ha = has_allow & S;   // check.go:416-425
        const u32 spm = (uload(&t.scope_flags[si]) >> 2) & 3u;
        if (spm == SP_REQUIRE_CONSENT) has_allow &= ~ha;
        else if (spm == SP_OVERRIDE_PARENT && ha) {
          p_allow |= ha; S &= ~ha;
#pragma unroll
          for (u32 k = 0; k < NA; ++k) if ((ha >> k) & 1u) aux[k * CBH_BLOCK + c.tid] = si;
        }
      }
    }
  }
  p_done = p_allow | p_deny;   // a definitive principal-policy result ends the action (check.go:445-448)
  W2_DBG(cyc2 = __builtin_readcyclecounter();)   // the principal pass is over
  walks &= ~((W)p_done * REP);

  // ---- the resource walk (cbh_check_flat.h: merged climb, deepest scope first)
  S = walks;
  for (;;) {
    const bool active = cur != CBH_NONE && (PRE ? pre_climbs : (S != 0 || !exists));
    if (wave_ballot(active) == 0) break;
    const u32 g_si = wave_max_bits(cur, active, scope_bits);
    const u64 here = wave_ballot(active && cur == g_si);
    const u32 lead = first_lane(here);
    const u32 g_ver = wave_readlane(r_ver, lead), g_k = wave_readlane(kind, lead);
    const bool ing = active && cur == g_si && r_ver == g_ver && kind == g_k;
    W2_DBG(++dbg_rounds;)
    const bool go = wave_ballot(ing && (PRE ? walks : S) != 0) != 0;
    uint4 bucket; bucket.x = bucket.y = bucket.z = bucket.w = 0;
    const bool have_bucket = udir_find(t, CBH_B_RESOURCE, g_ver, g_k, g_si, bucket);
    exists = exists || (ing && have_bucket);
    // role policies at (version, scope)?  Their resource patterns also make the resource "exist" (index.go:966-997)
    uint4 rpres; rpres.x = rpres.y = 0;
    bool rolepol_here = false;
    u64 g_kb = 0;
    const u32 g_sf = uload(&t.scope_flags[g_si]);
    if (has_rolepol && (g_sf & CBH_SCOPE_F_ROLEPOL) && udir_find(t, CBH_B_RPRES, g_ver, g_si, 0, rpres)) {
      rolepol_here = true;
      if (t.nfa_words[DIM_KIND]) g_kb = wave_readlane64(kind_bits(), lead);
      bool m = false;
      for (u32 k = 0; k < rpres.y && !m; ++k) m = pat_match(uload(&t.pool[rpres.x + k]), g_k, g_kb);
      exists = exists || (ing && m);
    }
    u32 site_flags = 0;   // pre-pass: which kinds of sites the bucket holds (cbh_blob.h CBH_BS_*)
    if (PRE && have_bucket) {
      uint4 ex; ex.x = 0;
      if (udir_find(t, CBH_B_RESEXISTS, g_ver, g_k, g_si, ex)) site_flags = ex.x;
      if (b.n_gslots <= t.gslots_generic) site_flags &= ~(u32)(CBH_BS_ROW_OPEN | CBH_BS_DR_OPEN);   // their slots are not filed for this batch
    }
    if (go) {
      if (!PRE && ing && mydepth < max_depth) { if (chain8) chain_si8[mydepth * CBH_BLOCK + c.tid] = (u8)g_si; else chain_si[mydepth * CBH_BLOCK + c.tid] = g_si; }
      const W S_before = S;
      if (PRE && have_bucket && t.n_dr && (pre_edr || (site_flags & (CBH_BS_DR_GENERIC | CBH_BS_DR_OPEN)))) {
        // the scope's derived roles (check.go:237-282): their sites, and - for programs that read runtime.* - their value
        u64 m = 0;
        for (u32 d = bucket.z; d < bucket.z + bucket.w; ++d) {
          const TblDrx dx = uload_rec<TblDrx>(t.drx, d);
          const bool applies = ing && (((dx.rm_lo & lane_rs_lo) | (dx.rm_hi & lane_rs_hi)) != 0);
          if (wave_ballot(applies) == 0) continue;
          if ((dx.p0 >> 16) != CBH_GSLOT_NONE) (void)leafish(dx.p1, 0u, no_leaf, dx.p0 >> 16, applies);   // the definition's variables
          u32 lv = 1u;
          if (dx.cond != CBH_NONE) lv = leafish(dx.cond, dx.flags & 3u, dx.leaf, dx.p0 & 0xFFFFu, applies);
          if (applies && (lv & 1u)) m |= 1ull << dx.name;
        }
        if (pre_edr && ing) edr_scope = m;
      }
      if (want_edr) {
        // effective derived roles (check.go:237-282): the definitions of this scope's policy, for the requests a walk is
        // still going for; which of the chain positions count - the ones a LEGITIMATE walk reached - is known at the fold
        u64 m = 0; bool de = false, du = false, dw = false;
        if (have_bucket) {
          for (u32 d = bucket.z; d < bucket.z + bucket.w; ++d) {
            const TblDrx dx = uload_rec<TblDrx>(t.drx, d);
            const bool applies = ing && S != 0 && (((dx.rm_lo & lane_rs_lo) | (dx.rm_hi & lane_rs_hi)) != 0);   // parent roles x the request's roles (check.go:244)
            if (wave_ballot(applies) == 0) continue;
            u32 lv = 1u;
            if ((dx.p0 >> 16) != CBH_GSLOT_NONE) {   // the definition's variables (evaluateVariables, check.go:612-633)
              const u32 pv = leafish(dx.p1, 0u, no_leaf, dx.p0 >> 16, applies);
              if (applies) { de = de || (pv & 2u) != 0; dw = dw || (pv & 8u) != 0; }
            }
            if (dx.cond != CBH_NONE) lv = leafish(dx.cond, dx.flags & 3u, dx.leaf, dx.p0 & 0xFFFFu, applies);
            if (applies) { if (lv & 1u) m |= 1ull << dx.name; de = de || (lv & 2u) != 0; du = du || (lv & 8u) != 0; }
          }
        }
        if (ing && S != 0) { edr_all |= m; derr_all = derr_all || de; dunsup_all = dunsup_all || du; dwtr_all = dwtr_all || dw; edr_vis = mydepth + 1u; }
      }
      if (rolepol_here) {
        // ---- synthetic DENYs of the role policies at this scope (index.go:352-530)
        uint4 ux; ux.y = ux.z = ux.w = 0;
        (void)udir_find(t, CBH_B_RESEXISTS, g_ver, g_k, g_si, ux);   // union of the bucket's role masks: Index.Query's base test
        u32 base = ing ? match_roles(ux.y, ux.z, ux.w & 0xFFFFu) : 0u;
        uint4 rl; rl.x = rl.y = 0;
        (void)udir_find(t, CBH_B_RPROLES, g_ver, g_si, 0, rl);
        const u32 n_rp = rl.y < CBH_W2_MAX_RP_ROLES ? rl.y : CBH_W2_MAX_RP_ROLES;
        // ... or a rule of the role policy of a role in the slot's set names the resource
        for (u32 j = 0; j < n_rp; ++j) {
          const u32 g_sr = uload(&t.pool[rl.x + j]);
          const u32 gc = (uload((const CBH_G u32*)(t.role_class + (g_sr & ~3u))) >> (8u * (g_sr & 3u))) & 0xFFu;   // its class (the lowering checks it has one)
          const u64 gm = 1ull << (gc < 62u ? gc : 63u);
          const u32 slot = ing ? match_roles((u32)gm, (u32)(gm >> 32), 0u) : 0u;
          if (wave_ballot(slot != 0) == 0) continue;
          uint4 rp;
          if (!udir_find(t, CBH_B_ROLEPOL, g_ver, g_si, g_sr, rp)) continue;
          bool any_res = false;
          for (u32 row = rp.x; row < rp.x + rp.y && !any_res; ++row) any_res = pat_match(uload(&t.rpx[(size_t)row * CBH_RPX_NF + CBH_RPX_RESOURCE]), g_k, g_kb);
          if (any_res) base |= slot;
        }
        // a slot's own role first (phase 0), then its ancestors in name order (phase 1): index.go:352-530 walks [role] ++ ancestors
        for (u32 jj = 0; jj < 2u * n_rp; ++jj) {
          const u32 j = jj < n_rp ? jj : jj - n_rp;
          const u32 g_sr = uload(&t.pool[rl.x + j]);
          const u32 gc = (uload((const CBH_G u32*)(t.role_class + (g_sr & ~3u))) >> (8u * (g_sr & 3u))) & 0xFFu;
          const u64 gm = 1ull << (gc < 62u ? gc : 63u);
          u32 own = 0;
#pragma unroll
          for (u32 k = 0; k < NR; ++k) own |= (u32)(k < role_cnt && rcls[k] == gc) << k;
          const u32 slot = ing ? (match_roles((u32)gm, (u32)(gm >> 32), 0u) & base & (jj < n_rp ? own : ~own)) : 0u;
          const W Wg = w2_rep_role<NA, NR>(slot) & (PRE ? walks : S);
          if (wave_ballot(Wg != 0) == 0) continue;
          uint4 rp;
          if (!udir_find(t, CBH_B_ROLEPOL, g_ver, g_si, g_sr, rp)) continue;
          u32 any_mask = 0, out_only = 0;   // actions some rule for this resource allows (subject to conditions); ... an output-only rule that shares its key
          for (u32 row = rp.x; row < rp.x + rp.y; ++row) {
            const TblRpx rr = uload_rec<TblRpx>(t.rpx, row);
            if (!pat_match(rr.resource, g_k, g_kb)) continue;
            const u32 ma = match_actions(rr.am_lo, rr.am_hi, rr.ag);
            any_mask |= ma;
            if ((rr.cnt & (CBH_RP_F_OUTPUT_ONLY | CBH_RP_F_SHARES_KEY)) == (CBH_RP_F_OUTPUT_ONLY | CBH_RP_F_SHARES_KEY)) out_only |= ma;
          }
          W dn = Wg & ~((W)any_mask * REP);   // no rule for the resource, or no allow action matched (index.go:436-461)
          if (EP && marks && dn != 0) mark(rp.z & 0x0FFFFFFFu);   // the synthetic DENY is a binding of the role policy
          for (u32 row = rp.x; row < rp.x + rp.y; ++row) {
            const TblRpx rr = uload_rec<TblRpx>(t.rpx, row);
            // (the reference visits a matched rule only if it has a condition - as the synthetic DENY row - or outputs)
            if ((rr.cond == CBH_NONE && !(rr.how & 4u)) || !pat_match(rr.resource, g_k, g_kb)) continue;
            const u32 ma = match_actions(rr.am_lo, rr.am_hi, rr.ag);
            const W mm = ((W)ma * REP) & Wg & ~dn;
            if (wave_ballot(mm != 0) == 0) continue;
            if (EP && marks && mm != 0) mark(rp.z & 0x0FFFFFFFu);
            if (rr.how & 4u) wtr |= mm;
            if (rp.w != CBH_NONE) {   // the policy's variables
              const u32 pv = leafish(PRE ? uload(&t.pool[rp.w]) : 0u, 0u, no_leaf, uload(&t.pool[rp.w + 1u]) & 0xFFFFu, mm != 0);
              err |= (pv & 2u) ? mm : 0u; wtr |= (pv & 8u) ? mm : 0u;
            }
            if (rr.cond == CBH_NONE) continue;
            // an action at or behind the first one an output-only rule of the same key is visited for finds "satisfied"
            // cached (check.go:324): the synthetic DENY would fire whatever the condition says (cbh_blob.h CBH_RP_F_*)
            if ((rr.cnt & CBH_RP_F_SHARES_KEY) && out_only != 0) unsup |= mm & ~((W)(((out_only & (0u - out_only)) - 1u) & AMASK) * REP);
            const u32 lv = leafish(rr.cond, rr.how & 3u, rr.leaf, rr.gslot & 0xFFFFu, mm != 0);
            err |= (lv & 2u) ? mm : 0u; unsup |= (lv & 8u) ? mm : 0u;
            if (!(lv & 1u)) dn |= mm;   // the synthetic row = DENY if none(condition)
          }
          if (!PRE) {
            dn &= S;
            if (dn) {   // check.go:395-403; the policy named is the role policy's
              deny |= dn; S &= ~dn;
              const u32 note = rp.z & 0x0FFFFFFFu;
#pragma unroll
              for (u32 k = 0; k < NA; ++k) {
                const W dk = (dn >> k) & REP;
                if (dk) {
                  const u32 r = w2_ctz(dk) / NA;
                  const u32 old = aux[k * CBH_BLOCK + c.tid];
                  if (old == CBH_NONE || r < (old >> 28)) aux[k * CBH_BLOCK + c.tid] = (r << 28) | note;
                }
              }
            }
          }
        }
      }
      if (have_bucket && bucket.y && (!PRE || (site_flags & (CBH_BS_ROW_GENERIC | CBH_BS_ROW_OPEN)))) {
        const u32 last = bucket.x + bucket.y - 1u;
        TblRowFull nxt = uload_rec<TblRowFull>(t.rows, bucket.x);
        for (u32 row = bucket.x; row <= last; ++row) {   // bindings in order (check.go:295-414)
          const TblRowFull rf = nxt;
          nxt = uload_rec<TblRowFull>(t.rows, row < last ? row + 1u : last);
          const TblRow& rw = rf.hot;
          W2_DBG(++dbg_rows;)
          u32 rm_lo = rw.rm_lo, rm_hi = rw.rm_hi, am_lo = rw.am_lo, am_hi = rw.am_hi, ag = 0, rg = 0, gslots = 0xFFFFFFFFu, probes = 0xFFFFFFFFu, probe_pcs = CBH_NONE;
          if (PRE && !(rw.flags & CBH_ROW_F_X)) continue;   // no site on this record
          if (rw.flags & CBH_ROW_F_X) {
            const TblRowX rx = uload_rec<TblRowX>(t.rowx, row);
            gslots = rx.gslots; probes = rx.p0; probe_pcs = rx.p1;
            if (PRE && (gslots & 0xFFFFu) >= b.n_gslots && (gslots >> 16) >= b.n_gslots && (probes & 0xFFFFu) >= b.n_gslots && (probes >> 16) >= b.n_gslots) continue;   // none filed for this batch
            if (rx.globs) { ag = rx.globs & 0xFFFFu; rg = rx.globs >> 16; rm_lo = rx.rm_lo; rm_hi = rx.rm_hi; am_lo = rx.am_lo; am_hi = rx.am_hi; }
          }
          // (a list with a glob is never skipped by class: the lanes' glob bits are still on their way from memory)
          if ((((rm_lo & wave_rc_lo) | (rm_hi & wave_rc_hi)) == 0 && rg == 0) || (((am_lo & wave_ac_lo) | (am_hi & wave_ac_hi)) == 0 && ag == 0)) continue;
          const u32 mact = match_actions(am_lo, am_hi, ag);
          const u32 mrole = match_roles(rm_lo, rm_hi, rg);
          // (the walks a visit is for: those still going - the pre-pass evaluates for every walk the request has)
          const W m = ing ? (w2_rep_role<NA, NR>(mrole) & ((W)mact * REP) & (PRE ? walks : S)) : (W)0;
          if (wave_ballot(m != 0) == 0) continue;
          if (EP && marks && m != 0) mark(rw.policy);   // the binding is iterated: its policy set is in effect (check.go:302-304)
          W hit = m;
          if (rw.flags & CBH_ROW_F_OUTPUT) wtr |= m;
          if (probes != 0xFFFFFFFFu) {   // the variables of the rule's policy are evaluated on every visit (check.go:306-321)
            const u32 pv = probe(probes & 0xFFFFu, probe_pcs, 0u, m != 0);
            err |= (pv & 2u) ? m : 0u; wtr |= (pv & 8u) ? m : 0u;
            if (rw.drcond != CBH_NONE) {   // ... the derived role's with its condition (check.go:328-334)
              const u32 pd = probe(probes >> 16, probe_pcs, 1u, m != 0);
              err |= (pd & 2u) ? m : 0u; wtr |= (pd & 8u) ? m : 0u;
            }
          }
          if (rw.drcond != CBH_NONE) {   // the derived-role condition first, the rule's own where that held (check.go:328-380)
            const u32 how = (rw.flags & CBH_ROW_F_DRLEAF_EMBEDDED) ? 1u : (rw.flags & CBH_ROW_F_DRTREE_EMBEDDED) ? 2u : 0u;
            const LeafRec l2 = uload_rec<LeafRec>(t.rowleaf2, row);
            const u32 lv = leafish(rw.drcond, how, l2, gslots >> 16, hit != 0);
            err |= (lv & 2u) ? hit : 0u; unsup |= (lv & 8u) ? hit : 0u;
            hit = (lv & 1u) ? hit : 0u;
          }
          if (rw.cond != CBH_NONE && wave_ballot(hit != 0) != 0) {
            const u32 how = (rw.flags & CBH_ROW_F_LEAF_EMBEDDED) ? 1u : (rw.flags & CBH_ROW_F_TREE_EMBEDDED) ? 2u : 0u;
            const u32 lv = leafish(rw.cond, how, rf.leaf, gslots & 0xFFFFu, hit != 0);
            err |= (lv & 2u) ? hit : 0u; unsup |= (lv & 8u) ? hit : 0u;
            hit = (lv & 1u) ? hit : 0u;
          }
          if (PRE) continue;
          if ((rw.flags & 3u) == CBH_EFFECT_ALLOW) has_allow |= hit;
          else if ((rw.flags & 3u) == CBH_EFFECT_DENY) { deny |= hit; S &= ~hit; }
        }
      }
      if (!PRE) {
        const W ha = ing ? (has_allow & S) : (W)0;   // check.go:416-425
        const u32 spm = (g_sf >> 2) & 3u;
        if (spm == SP_REQUIRE_CONSENT) has_allow &= ~ha;
        else if (spm == SP_OVERRIDE_PARENT) { allow |= ha; S &= ~ha; }
        const W newly = S_before & ~S;
        dp0 |= (mydepth & 1u) ? newly : (W)0; dp1 |= (mydepth & 2u) ? newly : (W)0; dp2 |= (mydepth & 4u) ? newly : (W)0; dp3 |= (mydepth & 8u) ? newly : (W)0;
      }
    }
    const u32 up = uchain_next(t, uload(&t.scope_parent[g_si]), FLAG_RES);
    if (ing) { cur = (mydepth + 1u < max_depth) ? up : CBH_NONE; ++mydepth; }
  }
  }   // (pass)

  W2_DBG(const u64 cyc3 = __builtin_readcyclecounter();)   // the walk is over
  if (PRE) {   // file the results of this lane's sites
    if (valid) for (u32 w = 0; w < b.n_gwords; ++w) b.gres[(size_t)w * NRQ + req] = gacc[w * CBH_BLOCK + c.tid];
#ifdef CBH_PROFILE_CYCLES
    if ((flags & CBH_F_DEBUG_CYCLES) && valid && act_cnt == 4 && o.policy && o.scope) {
      o.policy[act_off] = (u32)(cyc1 - cyc0); o.policy[act_off + 1] = (u32)(cyc2 - cyc1); o.policy[act_off + 2] = (u32)(cyc3 - cyc2); o.policy[act_off + 3] = (u32)cyc_eval;
      o.scope[act_off] = (u32)rt0; o.scope[act_off + 1] = (u32)__builtin_amdgcn_s_memrealtime(); o.scope[act_off + 2] = dbg_rows | (dbg_evals << 16); o.scope[act_off + 3] = dbg_rounds;
    }
#endif
    return;
  }

  // ---- nothing to evaluate at all?  (check.go:116-121, 165-170)
  const bool decided = nothing_to_evaluate();
  // what an action no rule decided reports (check.go:191, 216-225, 429-431)
  const u32 pol_none = decided ? ((u32)CBH_P_NO_MATCH << 28)
                     : role_cnt == 0 ? ((u32)CBH_P_EMPTY << 28)
                     : exists ? (((u32)CBH_P_RESOURCE << 28) | first) : ((u32)CBH_P_NO_MATCH << 28);
  const u32 pol_hit = ((u32)CBH_P_RESOURCE << 28) | first;
  if (decided) { p_allow = p_deny = p_err = p_unsup = p_wtr = 0; }

  // ---- the fold (check.go:429-442), per action: a principal policy's word, else the first role that allowed, else the
  // first role that denied
  u32 eff[(NA + 3u) / 4u] = {}, st[(NA + 3u) / 4u] = {}, pol[NA], scp[NA];
  u32 a_un = 0, a_er = 0, a_wt = 0;   // per action: outside the device subset / an evaluation error / something for the trace pass
#pragma unroll
  for (u32 k = 0; k < NA; ++k) {
    const W ak = (allow >> k) & REP, dk = (deny >> k) & REP;
    const W win = ak ? (ak & ((W)0 - ak)) : (dk & ((W)0 - dk));
    const W wb = win << k;
    const u32 d = ((dp0 & wb) ? 1u : 0u) | ((dp1 & wb) ? 2u : 0u) | ((dp2 & wb) ? 4u : 0u) | ((dp3 & wb) ? 8u : 0u);
    const bool pk = ((p_allow | p_deny) >> k) & 1u;
    const u32 note = aux[k * CBH_BLOCK + c.tid];
    u32 pw = win ? pol_hit : pol_none, sw = CBH_NONE;
    if (win && k < act_cnt) sw = chain8 ? (u32)chain_si8[d * CBH_BLOCK + c.tid] : chain_si[d * CBH_BLOCK + c.tid];
    if (win && !ak && note != CBH_NONE && (note >> 28) == w2_ctz(win) / NA) pw = ((u32)CBH_P_TABLE << 28) | (note & 0x0FFFFFFFu);   // a role policy denied
    bool al = ak != 0;
    if (pk) { pw = p_pol; sw = note; al = ((p_allow >> k) & 1u) != 0; }
    pol[k] = pw; scp[k] = sw;
    eff[k >> 2] |= (u32)(al ? CBH_EFFECT_ALLOW : CBH_EFFECT_DENY) << (8 * (k & 3u));   // NO_MATCH -> DENY (check.go:451-453)
    // an evaluation the reference would not have made - a role after the one that allowed - does not count
    const W seen = ak ? (((ak & ((W)0 - ak)) << 1) - 1u) : ~(W)0;
    const W ek = (err >> k) & REP & seen, uk = (unsup >> k) & REP & seen, wk = (wtr >> k) & REP & seen;
    a_er |= (u32)(ek != 0 || ((p_err >> k) & 1u)) << k;
    a_un |= (u32)(uk != 0 || ((p_unsup >> k) & 1u)) << k;
    a_wt |= (u32)(wk != 0 || ((p_wtr >> k) & 1u)) << k;
  }

  // ---- effective derived roles (check.go:237-282): the definitions of a scope's policy are evaluated when a role walk
  // REACHES that scope - a walk the reference really makes, i.e. not one of a role after the role that allowed its
  // action.  A decided walk reached the chain positions up to the one that decided it, an undecided one the whole chain;
  // the climb above left every position's roles in LDS.
  u64 edr = 0;
  if (want_edr) {
    W legit = 0;
#pragma unroll
    for (u32 k = 0; k < NA; ++k) {
      const W ak = (allow >> k) & REP;
      const W seen = ak ? (((ak & ((W)0 - ak)) << 1) - 1u) : ~(W)0;
      legit |= ((walks >> k) & REP & seen) << k;
    }
    const W done = allow | deny;
    u32 reach = 0;   // deepest chain position a legitimate walk reached, + 1 (0 = none)
    if (legit & ~done) reach = CBH_FLAT_MAX_DEPTH;
    else if (legit) {
      W cand = legit; u32 d = 0;
      W tp = cand & dp3; if (tp) { cand = tp; d |= 8u; }
      tp = cand & dp2; if (tp) { cand = tp; d |= 4u; }
      tp = cand & dp1; if (tp) { cand = tp; d |= 2u; }
      tp = cand & dp0; if (tp) { cand = tp; d |= 1u; }
      reach = d + 1u;
    }
    // The climb evaluated the definitions at every position some walk of the request was still going for.  That is
    // exactly the legitimate reach unless the last walks going were all of roles behind an allowing one - rare; those
    // requests (only) climb once more, for the definitions alone.
    edr = edr_all;
    bool derr = derr_all, dr_unsup = dunsup_all, dwtr = dwtr_all;
    const bool again = reach < edr_vis;
    if (wave_ballot(again) != 0) {
      if (again) { edr = 0; derr = false; dr_unsup = false; dwtr = false; }
      u32 cur2 = first, d2 = 0;
      for (;;) {
        const bool active = again && cur2 != CBH_NONE && d2 < reach;
        if (wave_ballot(active) == 0) break;
        const u32 g_si = wave_max_bits(cur2, active, scope_bits);
        const u32 lead = first_lane(wave_ballot(active && cur2 == g_si));
        const u32 g_ver = wave_readlane(r_ver, lead), g_k = wave_readlane(kind, lead);
        const bool ing = active && cur2 == g_si && r_ver == g_ver && kind == g_k;
        uint4 bucket; bucket.x = bucket.y = bucket.z = bucket.w = 0;
        if (udir_find(t, CBH_B_RESOURCE, g_ver, g_k, g_si, bucket)) {
          for (u32 d = bucket.z; d < bucket.z + bucket.w; ++d) {
            const TblDrx dx = uload_rec<TblDrx>(t.drx, d);
            const bool applies = ing && (((dx.rm_lo & lane_rs_lo) | (dx.rm_hi & lane_rs_hi)) != 0);
            if (wave_ballot(applies) == 0) continue;
            u32 lv = 1u;
            if ((dx.p0 >> 16) != CBH_GSLOT_NONE) {
              const u32 pv = leafish(dx.p1, 0u, no_leaf, dx.p0 >> 16, applies);
              if (applies) { derr = derr || (pv & 2u) != 0; dwtr = dwtr || (pv & 8u) != 0; }
            }
            if (dx.cond != CBH_NONE) lv = leafish(dx.cond, dx.flags & 3u, dx.leaf, dx.p0 & 0xFFFFu, applies);
            if (applies) { if (lv & 1u) edr |= 1ull << dx.name; derr = derr || (lv & 2u) != 0; dr_unsup = dr_unsup || (lv & 8u) != 0; }
          }
        }
        const u32 up = uchain_next(t, uload(&t.scope_parent[g_si]), FLAG_RES);
        if (ing) { cur2 = up; ++d2; }
      }
    }
    // evaluation errors are a per-request fact: every action of the request
    if (derr) a_er = AMASK;
    if (dr_unsup) a_un = AMASK;
    if (dwtr) a_wt = AMASK;
  }
#pragma unroll
  for (u32 k = 0; k < NA; ++k)
    st[k >> 2] |= (u32)(((a_un >> k) & 1u) ? CBH_ST_UNSUPPORTED : ((a_er >> k) & 1u) ? CBH_ST_CEL_ERROR : ((a_wt >> k) & 1u) ? CBH_ST_WANTS_TRACE : CBH_ST_OK) << (8 * (k & 3u));

#ifdef CBH_PROFILE_CYCLES
  if (flags & CBH_F_DEBUG_CYCLES) {
    const u64 cyc4 = __builtin_readcyclecounter();
#if CBH_PROFILE_CYCLES == 2   /* the prologue and the principal pass in parts */
    pol[0] = (u32)(cycA0 - cyc0); pol[1] = (u32)(cycA - cycA0); pol[2] = (u32)(cycB - cycA); pol[3] = (u32)(cycC - cycB);
    scp[0] = (u32)rt0; scp[1] = (u32)__builtin_amdgcn_s_memrealtime(); scp[2] = (u32)(cyc1 - cycC) | ((u32)((cycP - cyc1) >> 4) << 20); scp[3] = (u32)(cyc2 - cycP);
#else
    pol[0] = (u32)(cyc1 - cyc0); pol[1] = (u32)(cyc2 - cyc1); pol[2] = (u32)(cyc3 - cyc2); pol[3] = (u32)(cyc4 - cyc3);
    scp[0] = (u32)rt0; scp[1] = (u32)__builtin_amdgcn_s_memrealtime(); scp[2] = dbg_rows; scp[3] = dbg_rounds;
#endif
  }
#endif
  const bool packed = valid && act_cnt == 4 && (act_off & 3u) == 0;
  if (packed) {
#ifndef CBH_HOSTSIM
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#else
    struct u32x4 { u32 x, y, z, w; };
#endif
    if (o.edr) store_nt(o.edr + req, edr);
    store_nt((CBH_G u32*)(o.effect + act_off), eff[0]);
    if (o.status) store_nt((CBH_G u32*)(o.status + act_off), st[0]);
    if (o.policy) { u32x4 v; v.x = pol[0]; v.y = pol[1]; v.z = pol[2]; v.w = pol[3]; store_nt((CBH_G u32x4*)(o.policy + act_off), v); }
    if (o.scope) { u32x4 v; v.x = scp[0]; v.y = scp[1]; v.z = scp[2]; v.w = scp[3]; store_nt((CBH_G u32x4*)(o.scope + act_off), v); }
  } else if (valid) {
    if (o.edr) o.edr[req] = edr;
#pragma unroll
    for (u32 k = 0; k < NA; ++k) {
      if (k < act_cnt) {
        o.effect[act_off + k] = (u8)((eff[k >> 2] >> (8 * (k & 3u))) & 0xFFu);
        if (o.status) o.status[act_off + k] = (u8)((st[k >> 2] >> (8 * (k & 3u))) & 0xFFu);
        if (o.policy) o.policy[act_off + k] = pol[k];
        if (o.scope) o.scope[act_off + k] = scp[k];
      }
    }
  }
}

#ifndef CBH_HOSTSIM
#define CBH_W2_ATTRS __launch_bounds__(CBH_W2_THREADS)
#else
#define CBH_W2_ATTRS
#endif
// the walk: four independent waves to a workgroup, no evaluator call
template <u32 NA, u32 NR, bool EP = false>
__device__ __forceinline__ void w2_walk_kernel_body(const KernelArgs& a, const KernelArgs* __restrict__ ka) {
  const u32 ncc = a.t.inline_cols;   // no generic program runs here: only the columns the inline leaf code reads are parked in LDS
  const W2Layout ly = w2_layout(ncc, false, a.t.max_depth, a.t.n_scopes, false, 0, a.t.K, a.t.n_dr, NA, (a.flags & CBH_FI_PACKED_TAGS) != 0);
  Ctx c{a.t, a.b, a.now_ns, a.flags, threadIdx.x % CBH_BLOCK, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
        (CBH_L u32*)cbh_dyn_lds + (threadIdx.x / CBH_BLOCK) * ly.wave_dw, ncc, ka};
  w2_body<0, NA, NR, EP>(a, c, ly);
}
__global__ CBH_W2_ATTRS void cbh_walk2_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_walk_kernel_body<CBH_W2_NA, CBH_W2_NR>(a, ka); }
// the same walk for the requests with five to eight roles, and for those with nine to sixteen actions: 64-bit walk vectors
// (launched over the part of a batch that has any, CBH_FI_SKIP_WIDE)
__global__ CBH_W2_ATTRS void cbh_walk2_wide_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_walk_kernel_body<CBH_W2_NA, CBH_W2_WIDE_NR>(a, ka); }
__global__ CBH_W2_ATTRS void cbh_walk2_awide_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_walk_kernel_body<CBH_W2_AWIDE_NA, CBH_W2_NR>(a, ka); }
// ... keeping the audit trail (w2_body EP: the walk twice, the second one marks; cbh_check_batch_trail on a table this walk decides)
__global__ CBH_W2_ATTRS void cbh_walk2_trail_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_walk_kernel_body<CBH_W2_NA, CBH_W2_NR, true>(a, ka); }
__global__ CBH_W2_ATTRS void cbh_walk2_wide_trail_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_walk_kernel_body<CBH_W2_NA, CBH_W2_WIDE_NR, true>(a, ka); }
__global__ CBH_W2_ATTRS void cbh_walk2_awide_trail_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_walk_kernel_body<CBH_W2_AWIDE_NA, CBH_W2_NR, true>(a, ka); }
// the pre-pass: one wave to a workgroup, the shared evaluator with its operand stack (cbh_check_wave.h generic_kernel_body)
#if defined(CBH_PRE_WPE) && !defined(CBH_HOSTSIM)   /* lab: occupancy target of the pre-pass */
#define CBH_PRE_WAVES __attribute__((amdgpu_waves_per_eu(CBH_PRE_WPE, CBH_PRE_WPE)))
#else
#define CBH_PRE_WAVES
#endif
template <u32 NA, u32 NR>
__device__ __forceinline__ void w2_pre_kernel_body(const KernelArgs& a, const KernelArgs* __restrict__ ka) {
  __shared__ u64 s_val[CBH_STACK_DEPTH * CBH_BLOCK];
  __shared__ u64 l_val[CBH_MAX_LOCALS * CBH_BLOCK];
  __shared__ u64 it_cont[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u32 it_idx[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u32 it_state[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u8 s_tag[CBH_STACK_DEPTH * CBH_BLOCK];
  __shared__ u8 l_tag[CBH_MAX_LOCALS * CBH_BLOCK];
  {
    const u32 tid = threadIdx.x;
    for (u32 k = 0; k < CBH_STACK_DEPTH; ++k) { s_tag[k * CBH_BLOCK + tid] = CBH_T_ERR; s_val[k * CBH_BLOCK + tid] = 0; }
    for (u32 k = 0; k < CBH_MAX_LOCALS; ++k) { l_tag[k * CBH_BLOCK + tid] = CBH_T_ERR; l_val[k * CBH_BLOCK + tid] = 0; }
    for (u32 k = 0; k < CBH_MAX_ITERS; ++k) { it_cont[k * CBH_BLOCK + tid] = 0; it_idx[k * CBH_BLOCK + tid] = 0; it_state[k * CBH_BLOCK + tid] = 0; }
  }
  const u32 ncc = cached_columns(&a);
  const W2Layout ly = w2_layout(ncc, (a.t.flags & CBH_MF_NEEDS_ARENA) != 0, a.t.max_depth, a.t.n_scopes, true, a.b.n_gwords, a.t.K, a.t.n_dr, CBH_W2_NA, (a.flags & CBH_FI_PACKED_TAGS) != 0);
  Ctx c{a.t, a.b, a.now_ns, a.flags, threadIdx.x,
        (CBH_L u64*)s_val, (CBH_L u8*)s_tag, (CBH_L u64*)l_val, (CBH_L u8*)l_tag,
        (CBH_L u64*)it_cont, (CBH_L u32*)it_idx, (CBH_L u32*)it_state,
        (CBH_L u32*)cbh_dyn_lds, ncc, ka};
  w2_body<1, NA, NR>(a, c, ly);
}
__global__ __launch_bounds__(CBH_BLOCK) CBH_PRE_WAVES void cbh_walk2_pre_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_pre_kernel_body<CBH_W2_NA, CBH_W2_NR>(a, ka); }
__global__ __launch_bounds__(CBH_BLOCK) CBH_PRE_WAVES void cbh_walk2_pre_wide_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_pre_kernel_body<CBH_W2_NA, CBH_W2_WIDE_NR>(a, ka); }
__global__ __launch_bounds__(CBH_BLOCK) CBH_PRE_WAVES void cbh_walk2_pre_awide_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_pre_kernel_body<CBH_W2_AWIDE_NA, CBH_W2_NR>(a, ka); }

// ---- the pre-pass in two kernels (CBH_PRE_SPLIT=1; measured in the round after this one: DESIGN §7): the collector walks as the
// pre-pass does but evaluates nothing - it needs the walk's registers, not the interpreter's - and the interpreter runs over the
// sites' lists, 64 (request, program) items to a wave, the items of a wave mostly one program (requests are grouped by route).
template <u32 NA, u32 NR>
__device__ __forceinline__ void w2_collect_kernel_body(const KernelArgs& a, const KernelArgs* __restrict__ ka) {
  const u32 ncc = a.t.inline_cols;
  const W2Layout ly = w2_layout(ncc, false, a.t.max_depth, a.t.n_scopes, true, a.b.n_gwords, a.t.K, a.t.n_dr, CBH_W2_NA, (a.flags & CBH_FI_PACKED_TAGS) != 0);
  Ctx c{a.t, a.b, a.now_ns, a.flags, threadIdx.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (CBH_L u32*)cbh_dyn_lds, ncc, ka};
  w2_body<2, NA, NR>(a, c, ly);
}
__global__ __launch_bounds__(CBH_BLOCK) void cbh_walk2_collect_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) { w2_collect_kernel_body<CBH_W2_NA, CBH_W2_NR>(a, ka); }
// grid = slots x ceil(requests / 64); a block whose part of its slot's list is empty leaves at once
__global__ __launch_bounds__(CBH_BLOCK) void cbh_walk2_interp_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  __shared__ u64 s_val[CBH_STACK_DEPTH * CBH_BLOCK];
  __shared__ u64 l_val[CBH_MAX_LOCALS * CBH_BLOCK];
  __shared__ u64 it_cont[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u32 it_idx[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u32 it_state[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u8 s_tag[CBH_STACK_DEPTH * CBH_BLOCK];
  __shared__ u8 l_tag[CBH_MAX_LOCALS * CBH_BLOCK];
  const BatchDev& b = a.b;
  const u32 tid = threadIdx.x;
  const u32 n = b.req_hi - b.req_lo, chunks = (n + CBH_BLOCK - 1u) / CBH_BLOCK;
  const u32 slot = blockIdx.x / chunks, first = (blockIdx.x % chunks) * CBH_BLOCK;
  u32 cnt = uload(&b.site_cnt[slot]);
  if (cnt > b.site_cap) cnt = b.site_cap;
  if (first >= cnt) return;
  {
    for (u32 k = 0; k < CBH_STACK_DEPTH; ++k) { s_tag[k * CBH_BLOCK + tid] = CBH_T_ERR; s_val[k * CBH_BLOCK + tid] = 0; }
    for (u32 k = 0; k < CBH_MAX_LOCALS; ++k) { l_tag[k * CBH_BLOCK + tid] = CBH_T_ERR; l_val[k * CBH_BLOCK + tid] = 0; }
    for (u32 k = 0; k < CBH_MAX_ITERS; ++k) { it_cont[k * CBH_BLOCK + tid] = 0; it_idx[k * CBH_BLOCK + tid] = 0; it_state[k * CBH_BLOCK + tid] = 0; }
  }
  const bool valid = first + tid < cnt;
  const u64 item = b.site_list[(size_t)slot * b.site_cap + (valid ? first + tid : first)];
  const u32 req = (u32)item, ref = (u32)(item >> 32);
  const u32 ncc = cached_columns(&a);
  Ctx c{a.t, a.b, a.now_ns, a.flags, tid, (CBH_L u64*)s_val, (CBH_L u8*)s_tag, (CBH_L u64*)l_val, (CBH_L u8*)l_tag,
        (CBH_L u64*)it_cont, (CBH_L u32*)it_idx, (CBH_L u32*)it_state, (CBH_L u32*)cbh_dyn_lds, ncc, ka};
  fill_column_cache(c, b, b.n_requests, req);
  bool pend = valid;
  for (;;) {   // the programs of the wave's items, one after the other (mostly one)
    const u64 rem = wave_ballot(pend);
    if (rem == 0) break;
    const u32 g_ref = wave_readlane(ref, first_lane(rem));
    const bool mine = pend && ref == g_ref;
    pend = pend && !mine;
    const u32 r = eval_ref<true>(c.ka_mem, lds_of(c), req, 0, false, g_ref, mine);
    if (mine) {
      const u32 st = r >> 8;
      const u64 code = (u64)(((r & 0xFFu) == 1u ? 1u : 0u) | ((st & CBH_ST_CEL_ERROR) ? 2u : 0u) | ((st & CBH_ST_UNSUPPORTED) ? 8u : 0u));
      CBH_G u64* w = b.gres + (size_t)(slot / CBH_W2_SLOTS_PER_WORD) * b.n_requests + req;
#ifdef CBH_HOSTSIM
      *w |= code << (4u * (slot % CBH_W2_SLOTS_PER_WORD));
#else
      atomicOr((unsigned long long*)w, (unsigned long long)(code << (4u * (slot % CBH_W2_SLOTS_PER_WORD))));
#endif
    }
  }
}

// Does cbh_walk2_kernel decide this table's batches?  (CBH_MF_WALK2; not strict mode, whose immediate DENYs are order
// dependent.)  Requests with five to eight roles or nine to sixteen actions take the walk's wider forms, what is wider
// still is left to the general walk, lane by lane (CBH_FI_*, cbh_w2_class).
static inline bool cbh_walk2_applies(u32 table_flags, u32 eval_flags) {
  return (table_flags & CBH_MF_WALK2) && !(eval_flags & CBH_F_STRICT_EVALUATION);
}

// ---- which kernels decide a batch, and with how much dynamic LDS (shared by cbh_engine.hip and the host simulation)
struct CbhPlan {
  int kind;                      // 0 the general walk (cbh_check_wave.h), 1 a flat kernel, 2 cbh_walk2_kernel (+ its pre-pass when n_gwords)
  cbh_check_kernel_fn kernel;
  u32 threads;                   // workgroup size of `kernel`
  u32 n_gwords;                  // kind 2: 64-bit words of evaluation-site results per request (0 = no pre-pass)
  u32 n_gslots;                  // ... the sites filed: slots 0 .. n - 1 (the generic ones only for a batch of plain values)
  cbh_check_kernel_fn wide_kernel;   // kind 2, batch with requests wider than the walk's shapes: the general walk's kernel for those (else null)
  bool walk_wide;                    // kind 2, batch with requests of five to eight roles: cbh_walk2_wide_kernel (+ its pre-pass) for those
  bool walk_awide;                   // kind 2, batch with requests of nine to sixteen actions: cbh_walk2_awide_kernel (+ its pre-pass) for those
  bool trail;                        // kind 2: the walks are the cbh_walk2_*trail_kernel forms (CBH_F_WANT_EFFECTIVE_POLICIES)
};
static inline CbhPlan cbh_plan(u32 table_flags, u32 n_derived_roles, bool has_globs, u32 gslots_generic, u32 gslots_all, u32 max_actions,
                               u32 max_roles, bool plain_tags, u32 eval_flags, bool no_flat, bool no_walk2, u32 max_bucket, bool no_walk2_wide = false, bool masks = false) {
  CbhPlan p; p.n_gwords = 0; p.n_gslots = 0; p.wide_kernel = nullptr; p.walk_wide = false; p.walk_awide = false; p.trail = false;
  bool flat = false;
  // the effective policies of a call are what the reference's loops TOUCH, role by role in order (check.go:208-442, 302-304): the
  // general walk keeps that order; the flat kernels and cbh_walk2_kernel walk a request's roles side by side
  const bool want_ep = (eval_flags & CBH_F_WANT_EFFECTIVE_POLICIES) != 0;
  if (want_ep) {
    // a flat table's walks keep what they touched per chain position and sort it out at the fold (flat_body EP); a table of
    // cbh_walk2_kernel's walks twice (w2_body EP, below); every other table takes the general walk, which iterates roles and
    // bindings in the reference's own order
    p.kernel = cbh_pick_kernel(no_flat ? (table_flags & ~(u32)CBH_MF_FLAT) : table_flags, n_derived_roles, has_globs, max_actions, max_roles, plain_tags, eval_flags, max_bucket, &p.threads, &flat, masks);
    if (flat) { p.kind = 1; p.kernel = cbh_flat_trail_variant(p.kernel); return p; }
    p.kind = 0; p.kernel = cbh_check_trail_kernel; p.threads = CBH_BLOCK;
    if (no_walk2 || !cbh_walk2_applies(table_flags, eval_flags)) return p;
  } else
  p.kernel = cbh_pick_kernel(no_flat ? (table_flags & ~(u32)CBH_MF_FLAT) : table_flags, n_derived_roles, has_globs, max_actions, max_roles, plain_tags, eval_flags, max_bucket, &p.threads, &flat, masks);
  p.kind = flat ? 1 : 0;
  if (!flat && !no_walk2 && cbh_walk2_applies(table_flags, eval_flags)) {
    // (the batch's maxima only: a launch that finds no request of its class costs a few idle waves)
    p.walk_wide = max_roles > CBH_W2_NR && !no_walk2_wide;
    p.walk_awide = max_actions > CBH_W2_NA && !no_walk2_wide;
    const bool beyond_base = max_actions > CBH_W2_NA || max_roles > CBH_W2_NR;
    const bool beyond_all = max_actions > CBH_W2_AWIDE_NA || max_roles > CBH_W2_WIDE_NR || (max_actions > CBH_W2_NA && max_roles > CBH_W2_NR);
    if (no_walk2_wide ? beyond_base : beyond_all) p.wide_kernel = p.kernel;
    p.kind = 2; p.kernel = want_ep ? cbh_walk2_trail_kernel : cbh_walk2_kernel; p.threads = CBH_W2_THREADS; p.trail = want_ep;
    p.n_gwords = w2_gwords(gslots_generic, gslots_all, plain_tags);
    p.n_gslots = plain_tags ? gslots_generic : gslots_all;
  }
  return p;
}
// dynamic LDS of a one-wave workgroup of the general walk: the column cache and, for a table whose programs build lists, the arena
static inline size_t cbh_general_lds(u32 table_flags, u32 n_columns, bool packed_tags = false) {
  const u32 ncc = n_columns < CBH_CACHE_COLS ? n_columns : CBH_CACHE_COLS;
  return (size_t)CBH_CC_DWORDS(ncc, packed_tags) * 4 + ((table_flags & CBH_MF_NEEDS_ARENA) ? (size_t)CBH_ARENA_ENTRIES * CBH_BLOCK * 9 : 0);
}
// dynamic LDS of a launch of `kernel` (pre = the pre-pass of kind 2)
static inline size_t cbh_plan_lds(const CbhPlan& p, u32 table_flags, u32 table_max_depth, u32 table_scopes, u32 table_strings, u32 n_columns, u32 inline_cols, u32 table_n_dr, bool pre, u32 na = CBH_W2_NA, bool packed_tags = false) {
  const u32 ncc = n_columns < CBH_CACHE_COLS ? n_columns : CBH_CACHE_COLS;
  if (p.kind == 2) return w2_lds_bytes(w2_layout(pre ? ncc : inline_cols, (table_flags & CBH_MF_NEEDS_ARENA) != 0, table_max_depth, table_scopes, pre, p.n_gwords, table_strings, table_n_dr, na, packed_tags), pre ? 1u : CBH_W2_WAVES);
  const size_t wave = cbh_general_lds(table_flags, n_columns, packed_tags);
  if (p.kind == 1) return (wave + cbh_flat_chain_bytes(table_max_depth, table_scopes)) * (p.threads / CBH_BLOCK) + cbh_flat_class_bytes(table_strings)
                          + (cbh_is_mask_kernel(p.kernel) ? cbh_flat_mask_bytes(p.threads) : 0)
                          + (cbh_is_flat_trail_kernel(p.kernel) ? cbh_flat_trail_bytes(p.threads, table_max_depth) : 0);
  return wave;
}
