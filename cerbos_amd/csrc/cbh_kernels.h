// Device code of libcerbos_hip.so: the glob-resolve kernel and the decision kernel.
// Included by cbh_engine.hip (the product) and, with the HIP keywords shimmed, by the
// test-only host simulation under tests/hostsim (functional checks without a GPU).
#pragma once
#include "cbh_vm.h"
#ifndef NFA_MAXW
#define NFA_MAXW 8
#endif

// ======================================================================== device code

__device__ __forceinline__ u32 hash4(u32 a, u32 b, u32 c, u32 d) {
  u32 h = a * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + b * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + c * 0xC2B2AE3Du;
  h = (h ^ (h >> 16)) + d * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  return h;
}

__device__ inline bool dir_find(const TableDev& t, u32 k0, u32 k1, u32 k2, u32 k3, uint4& v) {
  u32 i = hash4(k0, k1, k2, k3) & t.hash_mask;
  for (u32 probe = 0; probe <= t.hash_mask; ++probe) {
    const uint4* s = reinterpret_cast<const uint4*>(&t.hash[i]);
    uint4 k = s[0];
    if (k.x == CBH_NONE) return false;
    if (k.x == k0 && k.y == k1 && k.z == k2 && k.w == k3) { v = s[1]; return true; }
    i = (i + 1) & t.hash_mask;
  }
  return false;
}

__device__ __forceinline__ u64 gbits_of(const TableDev& t, const BatchDev& b, u32 dim, u32 sid) {
  if (sid < t.K) return t.gbits[(size_t)dim * t.K + sid];
  return b.gbits[(size_t)dim * b.n_strings + (sid - t.K)];
}
__device__ __forceinline__ bool pat_match(u32 pref, u32 sid, u64 bits) {
  return (pref & CBH_PAT_GLOB) ? ((bits >> (pref & 63u)) & 1ull) != 0 : pref == sid;
}

#define DIM_ACTION 0
#define DIM_ROLE 1
#define DIM_KIND 2
#define FLAG_RES 1u
#define FLAG_PRIN 2u
#define SP_OVERRIDE_PARENT 1u
#define SP_REQUIRE_CONSENT 2u
#define EFF_NO_MATCH 0u

__device__ __forceinline__ u32 chain_next(const TableDev& t, u32 si, u32 flagbit) {
  while (si != CBH_NONE && !(t.scope_flags[si] & flagbit)) si = t.scope_parent[si];
  return si;
}
// first scope of the chain for a request scope word (ruletable.go:848-882)
__device__ __forceinline__ u32 chain_first(const TableDev& t, u32 raw, u32 flagbit, bool lenient) {
  const u32 si = raw & ~CBH_SCOPE_EXACT;
  const bool exact = (raw & CBH_SCOPE_EXACT) != 0;
  if (!lenient && !(exact && (t.scope_flags[si] & flagbit))) return CBH_NONE;
  return chain_next(t, si, flagbit);
}

#ifndef CBH_HOSTSIM
// --- glob NFA ----------------------------------------------------------------------------
#ifndef NFA_MAXW
#define NFA_MAXW 8
#endif
__global__ __launch_bounds__(CBH_BLOCK) void cbh_resolve_globs_kernel(TableDev t, BatchDev b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* lds = reinterpret_cast<u64*>(smem);
  const u32 i = blockIdx.x * CBH_BLOCK + threadIdx.x;
  for (u32 dim = 0; dim < 3; ++dim) {
    const u32 nw = t.nfa_words[dim];
    if (nw == 0) continue;  // uniform
    // stage init | star | cls[256] | self[256] into LDS (coalesced 8-byte loads)
    const u32 total = (2 + 512) * nw;
    __syncthreads();
    for (u32 k = threadIdx.x; k < total; k += CBH_BLOCK) lds[k] = t.nfa[dim][k];
    __syncthreads();
    if (i >= b.n_strings) continue;
    if (!(b.str_flags[i] & (1u << dim))) { b.gbits[(size_t)dim * b.n_strings + i] = 0; continue; }
    const u64* init = lds; const u64* star = lds + nw; const u64* cls = lds + 2 * nw; const u64* self = cls + 256 * nw;
    u64 A[NFA_MAXW];
#pragma unroll
    for (int w = 0; w < NFA_MAXW; ++w) A[w] = (u32)w < nw ? init[w] : 0;
    const u32 o = b.str_off[i], n = b.str_off[i + 1] - o;
    for (u32 k = 0; k < n; ++k) {
      const u32 ch = b.str_bytes[o + k];
      u64 carry = 0;
#pragma unroll
      for (int w = 0; w < NFA_MAXW; ++w) {
        if ((u32)w < nw) {
          const u64 adv = A[w] & cls[ch * nw + w];
          const u64 stay = A[w] & self[ch * nw + w];
          A[w] = (adv << 1) | carry | stay;
          carry = adv >> 63;
        }
      }
      // epsilon closure over star positions (a star may match the empty string)
      for (int it = 0; it < 8; ++it) {
        u64 c2 = 0; bool changed = false;
#pragma unroll
        for (int w = 0; w < NFA_MAXW; ++w) {
          if ((u32)w < nw) {
            const u64 s = A[w] & star[w];
            const u64 nx = A[w] | (s << 1) | c2;
            c2 = s >> 63;
            changed |= nx != A[w];
            A[w] = nx;
          }
        }
        if (!changed) break;
      }
    }
    // accept table follows the transition arrays in global memory
    const u32* acc = reinterpret_cast<const u32*>(t.nfa[dim] + (size_t)(2 + 512) * nw);
    const u32 nacc = acc[0];
    u64 out = 0;
    for (u32 k = 0; k < nacc; ++k) {
      const u32 bit = acc[2 + 2 * k], gi = acc[2 + 2 * k + 1];
      u64 word = 0;
#pragma unroll
      for (int w = 0; w < NFA_MAXW; ++w) if ((u32)w == (bit >> 6)) word = A[w];
      if ((word >> (bit & 63u)) & 1ull) out |= 1ull << gi;
    }
    b.gbits[(size_t)dim * b.n_strings + i] = out;
  }
}

#endif  // CBH_HOSTSIM

// --- decision kernel ---------------------------------------------------------------------
struct RoleSet {   // [role] ++ ancestors(role) for the request's resource scope (index.go:716-742)
  u32 role; u32 par_off; u32 par_cnt; u64 gbits;   // gbits: OR of role-dimension glob bits over the set
};

__device__ __forceinline__ bool roleset_has(const TableDev& t, const RoleSet& rs, u32 pref) {
  if (pref & CBH_PAT_GLOB) return ((rs.gbits >> (pref & 63u)) & 1ull) != 0;
  if (pref == rs.role) return true;
  for (u32 k = 0; k < rs.par_cnt; ++k) if (t.pool[rs.par_off + k] == pref) return true;
  return false;
}

// Evaluate the derived roles imported by resource policy (kind, version, scope) for this request
// (check.go:237-279).  Returns the mask of activated roles; err=true if (strict mode) one failed.
__device__ inline u64 eval_derived_roles(const Ctx& c, Lane& L, u32 dr_begin, u32 dr_cnt, u32 role_off,
                                         u32 role_cnt, u32 scope_key, bool& err) {
  const TableDev& t = c.t;
  u64 mask = 0; err = false;
  for (u32 d = dr_begin; d < dr_begin + dr_cnt; ++d) {
    const u32 pcnt = t.dr[CBH_DR_PARENTS_CNT * t.n_dr + d];
    bool applies = pcnt == CBH_NONE;
    if (!applies) {
      const u32 poff = t.dr[CBH_DR_PARENTS_OFF * t.n_dr + d];
      for (u32 r = 0; r < role_cnt && !applies; ++r) {
        const u32 role = c.b.roles[role_off + r];
        uint4 pv; u32 aoff = 0, acnt = 0;
        if (scope_key != CBH_NONE && dir_find(t, CBH_B_PARENTS, scope_key, role, 0, pv)) { aoff = pv.x; acnt = pv.y; }
        for (u32 k = 0; k < pcnt && !applies; ++k) {
          const u32 want = t.pool[poff + k];
          applies = want == role;
          for (u32 a = 0; a < acnt && !applies; ++a) applies = t.pool[aoff + a] == want;
        }
      }
    }
    if (!applies) continue;
    const u32 cond = t.dr[CBH_DR_COND * t.n_dr + d];
    int r = 1;
    if (cond != CBH_NONE) r = run_program(c, L, cond);
    if (r == 2) { err = true; continue; }
    if (r == 1) mask |= 1ull << t.dr[CBH_DR_NAME * t.n_dr + d];
  }
  return mask;
}

__global__ __launch_bounds__(CBH_BLOCK) void cbh_check_kernel(TableDev t, BatchDev b, OutDev o, i64 now_ns, u32 flags) {
  __shared__ u64 s_val[CBH_STACK_DEPTH * CBH_BLOCK];
  __shared__ u64 l_val[CBH_MAX_LOCALS * CBH_BLOCK];
  __shared__ u64 it_cont[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u32 it_idx[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u32 it_state[CBH_MAX_ITERS * CBH_BLOCK];
  __shared__ u8 s_tag[CBH_STACK_DEPTH * CBH_BLOCK];
  __shared__ u8 l_tag[CBH_MAX_LOCALS * CBH_BLOCK];

  const u32 tup = blockIdx.x * CBH_BLOCK + threadIdx.x;
  if (tup >= b.n_tuples) return;
  Ctx c{t, b, now_ns, flags, threadIdx.x, s_val, s_tag, l_val, l_tag, it_cont, it_idx, it_state};

  const u32 req = b.tuple_req[tup];
  const u32 act = b.tuple_action[tup];
  const u32 NR = b.n_requests;
#define RQ(f) b.req_u32[(size_t)(f) * NR + req]
  const u32 pid = RQ(CBH_RQ_PRINCIPAL_ID);
  const u32 p_scope = RQ(CBH_RQ_P_SCOPE), p_ver = RQ(CBH_RQ_P_VERSION);
  const u32 kind = RQ(CBH_RQ_KIND);
  const u32 r_scope = RQ(CBH_RQ_R_SCOPE), r_ver = RQ(CBH_RQ_R_VERSION);
  const u32 role_off = RQ(CBH_RQ_ROLE_OFF), role_cnt = RQ(CBH_RQ_ROLE_CNT);
#undef RQ
  const bool lenient = (flags & CBH_F_LENIENT_SCOPE_SEARCH) != 0;
  const bool want_edr = (flags & CBH_F_WANT_DERIVED_ROLES) != 0 || (t.flags & CBH_MF_USES_RUNTIME_EDR) != 0;

  Lane L; L.req = req; L.edr = 0; L.status = 0; L.edr_err = false;

  u32 eff = EFF_NO_MATCH, pol = ((u32)CBH_P_NO_MATCH << 28), scp = CBH_NONE;
  const u64 act_bits = gbits_of(t, b, DIM_ACTION, act);
  const u64 kind_bits = gbits_of(t, b, DIM_KIND, kind);
  // parent roles are looked up with the request's own resource scope only (check.go:172,227)
  const u32 pr_scope_key = (r_scope & CBH_SCOPE_EXACT) ? (r_scope & ~CBH_SCOPE_EXACT) : CBH_NONE;

  const u32 p_first = chain_first(t, p_scope, FLAG_PRIN, lenient);
  const u32 r_first = chain_first(t, r_scope, FLAG_RES, lenient);

  bool decided = false;
  if (p_first == CBH_NONE && r_first == CBH_NONE) decided = true;  // check.go:119-121

  bool p_exists = false, r_exists = false;
  if (!decided) {
    uint4 v;
    for (u32 si = p_first; si != CBH_NONE && !p_exists; si = chain_next(t, t.scope_parent[si], FLAG_PRIN))
      p_exists = dir_find(t, CBH_B_PPEXISTS, p_ver, si, 0, v);                      // index.go:999-1021
    for (u32 si = r_first; si != CBH_NONE && !r_exists; si = chain_next(t, t.scope_parent[si], FLAG_RES)) {
      if (dir_find(t, CBH_B_RESEXISTS, r_ver, kind, si, v)) { r_exists = true; break; }   // index.go:966-997
      if (dir_find(t, CBH_B_RPRES, r_ver, si, 0, v))
        for (u32 k = 0; k < v.y && !r_exists; ++k) r_exists = pat_match(t.pool[v.x + k], kind, kind_bits);
    }
    if (!p_exists && !r_exists) decided = true;                                       // check.go:168-170
  }

  if (!decided) {
    pol = ((u32)CBH_P_EMPTY << 28);   // zero EffectInfo (check.go:191)
    bool action_done = false;
    for (u32 pt = 0; pt < 2 && !action_done; ++pt) {       // 0 = principal policies, 1 = resource policies
      const bool is_res = pt == 1;
      const u32 first = is_res ? r_first : p_first;
      const u32 flagbit = is_res ? FLAG_RES : FLAG_PRIN;
      const bool exists = is_res ? r_exists : p_exists;
      eff = EFF_NO_MATCH;                                   // check.go:206
      const u32 n_iter = is_res ? role_cnt : (role_cnt ? 1u : 0u);   // check.go:208-213
      for (u32 ri = 0; ri < n_iter && !action_done; ++ri) {
        RoleSet rs; rs.role = b.roles[role_off + ri]; rs.par_off = 0; rs.par_cnt = 0;
        rs.gbits = gbits_of(t, b, DIM_ROLE, rs.role);
        {
          uint4 pv;
          if (pr_scope_key != CBH_NONE && dir_find(t, CBH_B_PARENTS, pr_scope_key, rs.role, 0, pv)) {
            rs.par_off = pv.x; rs.par_cnt = pv.y;
            for (u32 k = 0; k < rs.par_cnt; ++k) rs.gbits |= t.gbits[(size_t)DIM_ROLE * t.K + t.pool[rs.par_off + k]];
          }
        }
        bool has_allow = false;
        u32 r_eff = EFF_NO_MATCH, r_scp = CBH_NONE;
        u32 r_pol = exists ? (((u32)(is_res ? CBH_P_RESOURCE : CBH_P_PRINCIPAL) << 28) | first)
                           : ((u32)CBH_P_NO_MATCH << 28);

        for (u32 si = first; si != CBH_NONE; si = chain_next(t, t.scope_parent[si], flagbit)) {
          uint4 bucket; bool have_bucket;
          if (is_res) {
            have_bucket = dir_find(t, CBH_B_RESOURCE, r_ver, kind, si, bucket);
            if (want_edr) {                                  // check.go:237-282
              bool derr = false; u64 m = 0;
              if (have_bucket && bucket.w) m = eval_derived_roles(c, L, bucket.z, bucket.w, role_off, role_cnt, pr_scope_key, derr);
              L.edr = m; L.edr_err = derr;
              if (o.edr && m) atomicOr(reinterpret_cast<unsigned long long*>(&o.edr[req]), (unsigned long long)m);
            }
          } else {
            have_bucket = dir_find(t, CBH_B_PRINCIPAL, r_ver, si, pid, bucket);   // resource version: check.go:294
          }
          if (r_eff != EFF_NO_MATCH) break;                  // check.go:284-286
          bool break_scopes = false;

          if (is_res) {
            // synthetic DENYs from role policies come first (index.go:318-322, 352-530)
            for (u32 k = 0; k <= rs.par_cnt && !break_scopes && !action_done; ++k) {
              const u32 srole = k == 0 ? rs.role : t.pool[rs.par_off + k - 1];
              uint4 rp;
              if (!dir_find(t, CBH_B_ROLEPOL, r_ver, si, srole, rp)) continue;
              bool any_action = false;
              for (u32 row = rp.x; row < rp.x + rp.y && !any_action; ++row) {
                if (!pat_match(t.rprows[CBH_RP_RESOURCE * t.n_rprows + row], kind, kind_bits)) continue;
                const u32 ao = t.rprows[CBH_RP_ALLOW_OFF * t.n_rprows + row], ac = t.rprows[CBH_RP_ALLOW_CNT * t.n_rprows + row];
                for (u32 a = 0; a < ac && !any_action; ++a) any_action = pat_match(t.pool[ao + a], act, act_bits);
              }
              bool deny = !any_action;   // no resource binding, or no allow-action matched (index.go:436-461)
              for (u32 row = rp.x; row < rp.x + rp.y && !deny && !action_done; ++row) {
                if (!pat_match(t.rprows[CBH_RP_RESOURCE * t.n_rprows + row], kind, kind_bits)) continue;
                const u32 cond = t.rprows[CBH_RP_COND * t.n_rprows + row];
                if (cond == CBH_NONE) continue;
                const u32 ao = t.rprows[CBH_RP_ALLOW_OFF * t.n_rprows + row], ac = t.rprows[CBH_RP_ALLOW_CNT * t.n_rprows + row];
                bool m = false;
                for (u32 a = 0; a < ac && !m; ++a) m = pat_match(t.pool[ao + a], act, act_bits);
                if (!m) continue;
                const int r = run_program(c, L, cond);       // synthetic row = DENY if none(cond)
                if (r == 2) { eff = CBH_EFFECT_DENY; pol = ((u32)CBH_P_TABLE << 28) | rp.z; scp = si; action_done = true; }
                else if (r == 0) deny = true;
              }
              if (deny && !action_done) {
                r_eff = CBH_EFFECT_DENY; r_scp = si; r_pol = ((u32)CBH_P_TABLE << 28) | rp.z;   // check.go:395-403
                break_scopes = true;
              }
            }
          }

          if (have_bucket && !break_scopes && !action_done) {
            for (u32 row = bucket.x; row < bucket.x + bucket.y; ++row) {
              if (!pat_match(t.rows[CBH_ROW_ACTION * t.n_rows + row], act, act_bits)) continue;
              if (is_res) { if (!roleset_has(t, rs, t.rows[CBH_ROW_ROLE * t.n_rows + row])) continue; }
              else if (!pat_match(t.rows[CBH_ROW_RESOURCE * t.n_rows + row], kind, kind_bits)) continue;
              const u32 drc = t.rows[CBH_ROW_DRCOND * t.n_rows + row];
              const u32 cnd = t.rows[CBH_ROW_COND * t.n_rows + row];
              int r = 1;
              if (drc != CBH_NONE) r = run_program(c, L, drc);           // check.go:328-366
              if (r == 1 && cnd != CBH_NONE) r = run_program(c, L, cnd); // check.go:368-380
              if (r == 2) {                                               // strict evaluation error
                eff = CBH_EFFECT_DENY; pol = ((u32)CBH_P_TABLE << 28) | t.rows[CBH_ROW_POLICY * t.n_rows + row]; scp = si;
                action_done = true; break;
              }
              if (r != 1) continue;
              const u32 e = t.rows[CBH_ROW_FLAGS * t.n_rows + row] & 3u;
              if (e == CBH_EFFECT_ALLOW) has_allow = true;
              else if (e == CBH_EFFECT_DENY) { r_eff = CBH_EFFECT_DENY; r_scp = si; break_scopes = true; break; }
            }
          }
          if (action_done || break_scopes) break;
          if (has_allow) {                                   // check.go:416-425
            const u32 sp = (t.scope_flags[si] >> 2) & 3u;
            if (sp == SP_REQUIRE_CONSENT) has_allow = false;
            else if (sp == SP_OVERRIDE_PARENT) { r_eff = CBH_EFFECT_ALLOW; r_scp = si; break; }
          }
        }
        if (action_done) break;
        if (eff == EFF_NO_MATCH) { eff = r_eff; pol = r_pol; scp = r_scp; }          // check.go:429-431
        if (r_eff == CBH_EFFECT_ALLOW) { eff = r_eff; pol = r_pol; scp = r_scp; break; }
        else if (r_eff == CBH_EFFECT_DENY && (pol >> 28) == CBH_P_NO_MATCH_SCOPE_PERMISSIONS &&
                 (r_pol >> 28) != CBH_P_NO_MATCH_SCOPE_PERMISSIONS) { eff = r_eff; pol = r_pol; scp = r_scp; }
      }
      if (eff == CBH_EFFECT_ALLOW || eff == CBH_EFFECT_DENY) break;   // check.go:445-448
    }
  }
  if (eff == EFF_NO_MATCH) eff = CBH_EFFECT_DENY;                     // check.go:451-453

  o.effect[tup] = (u8)eff;
  if (o.policy) o.policy[tup] = pol;
  if (o.scope) o.scope[tup] = scp;
  if (o.status) o.status[tup] = (u8)((L.status & CBH_ST_UNSUPPORTED) ? CBH_ST_UNSUPPORTED : (L.status & CBH_ST_CEL_ERROR));
}

