// cbh_check_flat_kernel - the decision kernel for FLAT tables: resource policies only (no derived roles, role
// policies, parent roles or principal policies), every condition a fused leaf or a tree of them, every rule record
// decided by its class masks (cbh_blob.h CBH_ROW_F_*_BY_CLASS) - and for batches with at most four actions and
// four roles per request, outside strict mode.  The host picks it (cbh_pick_check_kernel); everything else runs
// on the general walk of cbh_check_wave.h.  Same contract, same outputs, bit for bit.
//
// What is different from the general walk: the loops are turned inside out.  check.go:208-442 walks
// roles -> scopes -> bindings; the role walks of one action are independent of each other until the final fold
// (first ALLOW wins, else the first DENY seeds, check.go:429-442), and a binding's condition does not depend on
// the role.  So here a wave walks scopes -> records ONCE, and every lane carries the state of all its role walks
// side by side: bit (4 r + k) of a 16-bit vector = "role r's walk for action k".  One visit of a record serves
// every role of every lane - match, condition, DENY / ALLOW bookkeeping are the same few instructions whatever
// the number of roles - instead of one visit per role.  The scope a walk was decided at is kept as its depth in
// the request's scope chain (four bit planes); the fold at the end picks, per action, the first role that
// allowed, else the first that denied.
//
// Evaluation errors stay exact: the reference never evaluates the roles after the one that allowed an action
// (check.go:433-436), so an error raised for bit (r, k) only counts if no role before r allowed k - decided in
// the fold, where that is known.  (Strict mode turns an error into an immediate DENY attributed to the rule's
// policy, check.go:353-356: order dependent, hence left to the general walk.)
#pragma once
#include "cbh_check_wave.h"

#define CBH_FLAT_MAX_DEPTH 16   /* scope chain entries a request can have here (the lowering checks the table) */

// OR of a 64-bit value over the wave (single-wave workgroups: LDS atomics + barriers that cost nothing)
__device__ __forceinline__ u64 wave_or64(u64 v) {
  __shared__ unsigned long long acc;
  if (threadIdx.x == 0) acc = 0;
  __syncthreads();
  if (v) atomicOr(&acc, (unsigned long long)v);
  __syncthreads();
  const u64 r = acc;
  __syncthreads();
  return r;
}

__device__ __forceinline__ void flat_body(const KernelArgs& ka_regs, Ctx& c) {
  const TableDev& t = ka_regs.t;
  const BatchDev& b = ka_regs.b;
  const OutDev& o = ka_regs.o;
  const u32 flags = ka_regs.flags;
  const u32 rix = b.req_lo + blockIdx.x * CBH_BLOCK + threadIdx.x;
  const bool valid = rix < b.req_hi;
  const u32 req = valid ? rix : b.req_lo;
  const u32 NR = b.n_requests;
#define RQ(f) b.req_u32[(size_t)(f) * NR + req]
  const u32 pid = RQ(CBH_RQ_PRINCIPAL_ID), kind = RQ(CBH_RQ_KIND), r_scope = RQ(CBH_RQ_R_SCOPE), r_ver = RQ(CBH_RQ_R_VERSION);
  const u32 role_off = RQ(CBH_RQ_ROLE_OFF), act_off = RQ(CBH_RQ_ACT_OFF);
  const u32 role_cnt = valid ? RQ(CBH_RQ_ROLE_CNT) : 0, act_cnt = valid ? RQ(CBH_RQ_ACT_CNT) : 0;   // both <= 4 (host-checked)
#undef RQ
  fill_column_cache(c, b, NR, req);
  const u32 all = (1u << act_cnt) - 1u;
  // actions and roles -> classes (CBH_SEC_ACTION_CLASS / CBH_SEC_ROLE_CLASS; 63 = a string no rule names).
  // A flat table has fewer than 32 classes per dimension and its masks mirror "any other string" (bit 63)
  // in bit 31 of the low dword: the match is a 1-bit field extract from ONE dword at a per-lane position.
  u32 ac[4], rc[4];
#pragma unroll
  for (u32 k = 0; k < 4; ++k) {
    const u32 a = k < act_cnt ? b.tuple_action[act_off + k] : CBH_NONE;
    const u32 ca = a < t.K ? (u32)t.action_class[a] : 63u;
    ac[k] = ca < 31u ? ca : 31u;
    const u32 r = k < role_cnt ? b.roles[role_off + k] : CBH_NONE;
    const u32 cr = r < t.K ? (u32)t.role_class[r] : 63u;
    rc[k] = cr < 31u ? cr : 31u;
  }
  u32 lane_ac = 0, lane_rc = 0;
  u32 walks = 0;   // bit 4r + k: role r exists and action k exists
#pragma unroll
  for (u32 k = 0; k < 4; ++k) {
    if (k < act_cnt) lane_ac |= 1u << ac[k];
    if (k < role_cnt) { lane_rc |= 1u << rc[k]; walks |= all << (4 * k); }
  }
  // classes present in the wave: a record none of them can match is skipped on the scalar unit
  const u64 wave_cls = wave_or64((u64)lane_ac | ((u64)lane_rc << 32));
  const u32 wave_ac = (u32)wave_cls, wave_rc = (u32)(wave_cls >> 32);

  const bool lenient = (flags & CBH_F_LENIENT_SCOPE_SEARCH) != 0;
  Lane L; L.req = req; L.edr = 0; L.status = 0; L.edr_err = false; L.pid = pid;

  __shared__ u32 chain_si[CBH_FLAT_MAX_DEPTH * CBH_BLOCK];   // [depth][lane]: scope index at that depth of the lane's chain
  u32 S = walks;                 // walks still going
  u32 has_allow = 0, allow = 0, deny = 0, err = 0, unsup = 0;
  u32 dp0 = 0, dp1 = 0, dp2 = 0, dp3 = 0;   // bit planes of the depth a walk was decided at
  u32 first = CBH_NONE; bool exists = false;

  bool pend = true;   // every lane takes part in routing: "no policy at all" is an answer too (check.go:119-121, 168-170)
  for (;;) {   // ---- waterfall over groups that share (scope, version, kind); a sorted batch has one per wave
    const u64 rem = wave_ballot(pend);
    if (rem == 0) break;
    const u32 lead = first_lane(rem);
    const u32 g_rs = wave_readlane(r_scope, lead), g_ver = wave_readlane(r_ver, lead), g_k = wave_readlane(kind, lead);
    const bool ing = pend && r_scope == g_rs && r_ver == g_ver && kind == g_k;
    pend = pend && !ing;
    const u32 g_first = uchain_first(t, g_rs, FLAG_RES, lenient);
    bool g_exists = false;
    u32 depth = 0;
    for (u32 si = g_first; si != CBH_NONE && depth < CBH_FLAT_MAX_DEPTH; si = uchain_next(t, uload(&t.scope_parent[si]), FLAG_RES), ++depth) {   // check.go:231
      const bool go = wave_ballot(ing && S != 0) != 0;
      if (!go && g_exists) break;   // every walk of the group is decided and a policy is known to exist
      uint4 bucket; bucket.x = bucket.y = bucket.z = bucket.w = 0;
      const bool have_bucket = udir_find(t, CBH_B_RESOURCE, g_ver, g_k, si, bucket);   // present for every resource policy (index.go:966-997)
      g_exists = g_exists || have_bucket;
      if (!go) continue;
      if (ing) chain_si[depth * CBH_BLOCK + c.tid] = si;
      const u32 S_before = S;
      if (have_bucket) {
        for (u32 row = bucket.x; row < bucket.x + bucket.y; ++row) {   // bindings in order (check.go:295-414)
          const TblRow rw = uload_rec<TblRow>(t.rows, 2 * row);
          const LeafRec lf = uload_rec<LeafRec>(t.rows, 2 * row + 1);
          if ((rw.rm_lo & wave_rc) == 0 || (rw.am_lo & wave_ac) == 0) continue;
          const u32 mact = ((rw.am_lo >> ac[0]) & 1u) | (((rw.am_lo >> ac[1]) & 1u) << 1) | (((rw.am_lo >> ac[2]) & 1u) << 2) | (((rw.am_lo >> ac[3]) & 1u) << 3);
          // one nibble per role (sign-extended 1-bit extracts), one bit per nibble for the actions; walks of other groups sit out
          const u32 mrole = ((0u - ((rw.rm_lo >> rc[0]) & 1u)) & 0xFu) | ((0u - ((rw.rm_lo >> rc[1]) & 1u)) & 0xF0u) |
                            ((0u - ((rw.rm_lo >> rc[2]) & 1u)) & 0xF00u) | ((0u - ((rw.rm_lo >> rc[3]) & 1u)) & 0xF000u);
          const u32 m = ing ? (mrole & (mact * 0x1111u) & S) : 0u;
          if (wave_ballot(m != 0) == 0) continue;
          int r = 1;
          if (rw.cond != CBH_NONE) {
            // once per record and request, whatever the roles (check.go:316-340)
            r = (rw.flags & CBH_ROW_F_LEAF_EMBEDDED) ? eval_cond_rec<false>(c, L, rw.cond, lf, m != 0) : eval_cond<false>(c, L, rw.cond, m != 0);
            if (L.status & CBH_ST_CEL_ERROR) err |= m;
            if (L.status & CBH_ST_UNSUPPORTED) unsup |= m;
            L.status = 0;
          }
          if (m != 0 && r == 1) {
            if ((rw.flags & 3u) == CBH_EFFECT_ALLOW) has_allow |= m;
            else if ((rw.flags & 3u) == CBH_EFFECT_DENY) { deny |= m; S &= ~m; }   // ends these walks (check.go:392-403)
          }
        }
      }
      const u32 ha = ing ? (has_allow & S) : 0u;   // check.go:416-425
      const u32 sp = (uload(&t.scope_flags[si]) >> 2) & 3u;
      if (sp == SP_REQUIRE_CONSENT) has_allow &= ~ha;
      else if (sp == SP_OVERRIDE_PARENT) { allow |= ha; S &= ~ha; }
      const u32 newly = S_before & ~S;
      dp0 |= (depth & 1u) ? newly : 0u; dp1 |= (depth & 2u) ? newly : 0u; dp2 |= (depth & 4u) ? newly : 0u; dp3 |= (depth & 8u) ? newly : 0u;
    }
    if (ing) { first = g_first; exists = g_exists; S = 0; }
  }

  // ---- the fold (check.go:429-442), per action: the first role that allowed, else the first role that denied
  const bool decided = first == CBH_NONE || !exists;   // nothing to evaluate: "NO_MATCH" (check.go:119-121, 168-170)
  const u32 pol_none = (u32)(decided ? CBH_P_NO_MATCH : (role_cnt ? CBH_P_RESOURCE : CBH_P_EMPTY)) << 28 | ((!decided && role_cnt) ? first : 0u);
  const u32 pol_hit = ((u32)CBH_P_RESOURCE << 28) | first;
  u32 eff4 = 0, st4 = 0, pol[4], scp[4];
#pragma unroll
  for (u32 k = 0; k < 4; ++k) {
    const u32 ak = (allow >> k) & 0x1111u, dk = (deny >> k) & 0x1111u;
    const u32 win = ak ? (ak & (0u - ak)) : (dk & (0u - dk));   // lowest role bit of the deciding kind
    const u32 wb = win << k;                                     // back at its position 4r + k
    const u32 d = ((dp0 & wb) ? 1u : 0u) | ((dp1 & wb) ? 2u : 0u) | ((dp2 & wb) ? 4u : 0u) | ((dp3 & wb) ? 8u : 0u);
    pol[k] = win ? pol_hit : pol_none;
    scp[k] = CBH_NONE;
    if (win && k < act_cnt) scp[k] = chain_si[d * CBH_BLOCK + c.tid];
    eff4 |= (u32)(ak ? CBH_EFFECT_ALLOW : CBH_EFFECT_DENY) << (8 * k);   // NO_MATCH -> DENY (check.go:451-453)
    // an evaluation the reference would not have made - a role after the one that allowed - does not count
    const u32 seen = ak ? (((ak & (0u - ak)) << 1) - 1u) : 0xFFFFu;
    const u32 ek = (err >> k) & 0x1111u & seen, uk = (unsup >> k) & 0x1111u & seen;
    st4 |= (u32)(uk ? CBH_ST_UNSUPPORTED : (ek ? CBH_ST_CEL_ERROR : CBH_ST_OK)) << (8 * k);
  }

  const bool packed = valid && act_cnt == 4 && (act_off & 3u) == 0;
  if (packed) {
    struct __attribute__((aligned(16))) u32x4 { u32 x, y, z, w; };
    if (o.edr) o.edr[req] = 0;
    *(CBH_G u32*)(o.effect + act_off) = eff4;
    if (o.status) *(CBH_G u32*)(o.status + act_off) = st4;
    if (o.policy) { u32x4 v; v.x = pol[0]; v.y = pol[1]; v.z = pol[2]; v.w = pol[3]; *(CBH_G u32x4*)(o.policy + act_off) = v; }
    if (o.scope) { u32x4 v; v.x = scp[0]; v.y = scp[1]; v.z = scp[2]; v.w = scp[3]; *(CBH_G u32x4*)(o.scope + act_off) = v; }
  } else if (valid) {
    if (o.edr) o.edr[req] = 0;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
      if (k < act_cnt) {
        o.effect[act_off + k] = (u8)((eff4 >> (8 * k)) & 0xFFu);
        if (o.status) o.status[act_off + k] = (u8)((st4 >> (8 * k)) & 0xFFu);
        if (o.policy) o.policy[act_off + k] = pol[k];
        if (o.scope) o.scope[act_off + k] = scp[k];
      }
    }
  }
}

#ifndef CBH_HOSTSIM
#define CBH_FLAT_ATTRS __launch_bounds__(CBH_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
#else
#define CBH_FLAT_ATTRS
#endif
__global__ CBH_FLAT_ATTRS void cbh_check_flat_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  const u32 ncc = cached_columns(&a);
  Ctx c{a.t, a.b, a.now_ns, a.flags, threadIdx.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
        (CBH_L u32*)cbh_dyn_lds, ncc, ka};
  flat_body(a, c);
}

// Which kernel decides this batch: the flat one when table (CBH_MF_FLAT), batch shape (<= 4 actions and <= 4 roles per
// request) and evaluation mode (not strict) allow it, else the general walk's instantiation for the table class.
static inline cbh_check_kernel_fn cbh_pick_kernel(u32 table_flags, u32 n_derived_roles, bool has_globs, u32 max_actions, u32 max_roles, u32 eval_flags) {
  if ((table_flags & CBH_MF_FLAT) && max_actions <= 4 && max_roles <= 4 && !(eval_flags & CBH_F_STRICT_EVALUATION)) return cbh_check_flat_kernel;
  return cbh_pick_check_kernel(table_flags, n_derived_roles, has_globs, max_actions);
}
