// oracle/ccheck.cpp - TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
//
// A scalar C++ restatement of the reference's decision algorithm, one CheckInput at a time and one
// action at a time exactly as ruletable.(*RuleTable).check loops (internal/ruletable/check.go:97-460,
// SURVEY.md Appendix A), written against the same lowered table image and flattened batch the GPU
// library consumes (cerbos_amd/csrc/cbh_blob.h, include/cerbos_hip.h).  It exists for two jobs:
//   * bench.py's `cpu_baseline` leg ("kind": "port"): a compiled CPU figure next to the GPU one;
//   * full-size parity in tests/ (every tuple of a 1M / 4M-tuple batch, not a sample).
// It shares only data-format headers with the product, no evaluator code.  It is itself pinned by
// tests/test_ccheck.py against oracle/check.py (the restatement that is pinned on the reference's
// golden fixtures) on the golden store and the synthetic configurations.
//
// Covers principal and resource policies, scope chains and scope permissions, derived roles, role policies
// (the synthetic DENY bindings of index.go:352-530) and parent roles (index.go:716-788).
// Deliberately partial in one respect - it reports instead of guessing: conditions that are not a fused
// comparison or an all/any/none tree of them -> the request's tuples carry CBH_ST_UNSUPPORTED
// (oracle/check.py stays the oracle for those).
//
// Build: g++ -O2 -std=c++17 -shared -fPIC -pthread -Iinclude -Icerbos_amd/csrc oracle/ccheck.cpp -o oracle/libccheck.so
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "cbh_blob.h"
#include "cerbos_hip.h"

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;
typedef uint8_t u8;

namespace {

struct Img {
  const u8* base = nullptr;
  u32 meta[CBH_META_N];
  const u32 *str_off, *scope_parent, *scope_flags, *rows, *rowpat, *rprows, *pool, *dr, *code, *const_rec, *theap_rec;
  const u8* str_bytes;
  const CbhHashSlot* hash;
  const u64* gbits;
  const u64* nfa[3];
  u32 K, hash_mask, nfa_words[3];
};

const u8* section(const u8* blob, u32 id) {
  const CbhBlobHeader* h = (const CbhBlobHeader*)blob;
  const CbhBlobSection* s = (const CbhBlobSection*)(blob + sizeof(CbhBlobHeader));
  for (u32 i = 0; i < h->n_sections; ++i) if (s[i].id == id) return blob + s[i].offset;
  return nullptr;
}

bool parse(Img& g, const u8* blob, size_t len) {
  const CbhBlobHeader* h = (const CbhBlobHeader*)blob;
  if (len < sizeof(*h) || h->magic != CBH_BLOB_MAGIC || h->version != CBH_BLOB_VERSION || h->total_len != len) return false;
  g.base = blob;
  memcpy(g.meta, section(blob, CBH_SEC_META), sizeof(g.meta));
  g.str_off = (const u32*)section(blob, CBH_SEC_STR_OFF); g.str_bytes = section(blob, CBH_SEC_STR_BYTES);
  g.scope_parent = (const u32*)section(blob, CBH_SEC_SCOPE_PARENT); g.scope_flags = (const u32*)section(blob, CBH_SEC_SCOPE_FLAGS);
  g.hash = (const CbhHashSlot*)section(blob, CBH_SEC_HASH); g.hash_mask = g.meta[CBH_M_HASH_MASK];
  g.rows = (const u32*)section(blob, CBH_SEC_ROWS); g.rowpat = (const u32*)section(blob, CBH_SEC_ROWPAT); g.rprows = (const u32*)section(blob, CBH_SEC_RPROWS); g.pool = (const u32*)section(blob, CBH_SEC_U32POOL);
  g.dr = (const u32*)section(blob, CBH_SEC_DR); g.code = (const u32*)section(blob, CBH_SEC_CODE);
  g.const_rec = (const u32*)section(blob, CBH_SEC_CONST_REC); g.theap_rec = (const u32*)section(blob, CBH_SEC_THEAP_REC);
  g.gbits = (const u64*)section(blob, CBH_SEC_GBITS); g.K = g.meta[CBH_M_NSTRINGS];
  g.nfa[0] = (const u64*)section(blob, CBH_SEC_NFA_ACTION); g.nfa[1] = (const u64*)section(blob, CBH_SEC_NFA_ROLE);
  g.nfa[2] = (const u64*)section(blob, CBH_SEC_NFA_KIND);
  g.nfa_words[0] = g.meta[CBH_M_NFA_WORDS_ACTION]; g.nfa_words[1] = g.meta[CBH_M_NFA_WORDS_ROLE]; g.nfa_words[2] = g.meta[CBH_M_NFA_WORDS_KIND];
  return g.hash && g.code && g.str_off && g.scope_flags;
}

u32 hash4(u32 a, u32 b, u32 c, u32 d) {   // the directory's hash (cerbos_amd/lower/blob.py hash4)
  u32 h = a * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + b * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + c * 0xC2B2AE3Du;
  h = (h ^ (h >> 16)) + d * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  return h;
}
const CbhHashSlot* dir_find(const Img& g, u32 k0, u32 k1, u32 k2, u32 k3) {
  u32 i = hash4(k0, k1, k2, k3) & g.hash_mask;
  for (u32 p = 0; p <= g.hash_mask; ++p, i = (i + 1) & g.hash_mask) {
    const CbhHashSlot& s = g.hash[i];
    if (s.k0 == CBH_NONE) return nullptr;
    if (s.k0 == k0 && s.k1 == k1 && s.k2 == k2 && s.k3 == k3) return &s;
  }
  return nullptr;
}

struct Str { const u8* p; u32 n; };
Str str_of(const Img& g, const cbh_batch& b, u32 sid) {
  if (sid < g.K) return Str{g.str_bytes + g.str_off[sid], g.str_off[sid + 1] - g.str_off[sid]};
  const u32 j = sid - g.K;
  return Str{b.str_bytes + b.str_off[j], b.str_off[j + 1] - b.str_off[j]};
}

// glob match bits of a string in one dimension (glob_dimension.go:62-95): table strings carry them in
// the image, batch-local strings run the bit-parallel automaton of the dimension (layout: cbh_blob.h)
u64 glob_bits(const Img& g, const cbh_batch& b, u32 dim, u32 sid) {
  const u32 W = g.nfa_words[dim];
  if (W == 0) return 0;
  if (sid < g.K) return g.gbits[(size_t)dim * g.K + sid];
  const u64* n = g.nfa[dim];
  const u64 *init = n, *star = n + W, *cls = n + 2 * W, *self = n + 2 * W + 256 * (size_t)W;
  const u32* tail = (const u32*)(n + 2 * W + 512 * (size_t)W);
  std::vector<u64> A(init, init + W), T(W);
  auto closure = [&]() {
    for (;;) {
      bool ch = false; u64 carry = 0;
      for (u32 w = 0; w < W; ++w) {
        const u64 m = A[w] & star[w], sh = (m << 1) | carry;
        carry = m >> 63;
        if (sh & ~A[w]) { A[w] |= sh; ch = true; }
      }
      if (!ch) break;
    }
  };
  closure();
  const Str s = str_of(g, b, sid);
  for (u32 i = 0; i < s.n; ++i) {
    const u64 *c = cls + (size_t)s.p[i] * W, *sf = self + (size_t)s.p[i] * W;
    u64 carry = 0;
    for (u32 w = 0; w < W; ++w) {
      const u64 m = A[w] & c[w];
      T[w] = (m << 1) | carry | (A[w] & sf[w]);
      carry = m >> 63;
    }
    A = T;
    closure();
  }
  u64 bits = 0;
  for (u32 k = 0; k < tail[0]; ++k) {
    const u32 pos = tail[2 + 2 * k], gi = tail[3 + 2 * k];
    if ((A[pos >> 6] >> (pos & 63)) & 1) bits |= 1ull << gi;
  }
  return bits;
}
bool pat_match(u32 pref, u32 sid, u64 bits) { if (pref == CBH_PAT_ANY) return true; return (pref & CBH_PAT_GLOB) ? ((bits >> (pref & 63u)) & 1) != 0 : pref == sid; }

// ---- CEL values of the fused-leaf subset ---------------------------------------------------------
struct V { u32 t; u64 v; };
bool is_num(u32 t) { return t == CBH_T_INT || t == CBH_T_UINT || t == CBH_T_DOUBLE; }
double f64_of(u64 v) { double d; memcpy(&d, &v, 8); return d; }
// exact ordering across int64 / uint64 / double: -1, 0, 1 or 2 (unordered, NaN).  x86-64 long double
// has a 64-bit significand, so all three convert exactly.
int num_cmp(V a, V b) {
  auto ld = [](V x) -> long double {
    return x.t == CBH_T_INT ? (long double)(i64)x.v : x.t == CBH_T_UINT ? (long double)x.v : (long double)f64_of(x.v);
  };
  const long double p = ld(a), q = ld(b);
  if (p != p || q != q) return 2;
  return p < q ? -1 : p > q ? 1 : 0;
}

struct Req {
  const Img& g; const cbh_batch& b; u32 r;
  bool unsupported = false;
  V column(u32 c) const {
    const size_t ix = (size_t)c * b.n_requests + r;
    const u32 t = b.col_tag[ix];
    if (t == CBH_T_ABSENT) return V{CBH_T_ERR, 0};
    return V{t, b.col_val[ix]};
  }
  V heap(u32 sel, u32 idx) const {
    if (sel == CBH_HEAP_TABLE) { const u32* q = g.theap_rec + 4 * (size_t)idx; return V{q[0], (u64)q[2] | ((u64)q[3] << 32)}; }
    if (sel == CBH_HEAP_BATCH) return V{b.heap_tag[idx], b.heap_val[idx]};
    return V{CBH_T_STRING, b.roles[idx]};
  }
  V operand(u32 kind, u32 arg) const {
    if (kind == 0) { const u32* q = g.const_rec + 4 * (size_t)arg; return V{q[0], (u64)q[2] | ((u64)q[3] << 32)}; }
    if (kind == 1 || kind == 3) return column(arg);   // 3: a column (the GPU keeps the first ones in LDS)
    if (kind == 4) arg = CBH_RQ_PRINCIPAL_ID;         // 4: P.id
    return V{CBH_T_STRING, b.req_u32[(size_t)arg * b.n_requests + r]};
  }
  // equality as CEL defines it: 1 / 0, or 4 = outside this restatement's subset
  int equal(V x, V y) {
    if (x.t == y.t) {
      if (x.t == CBH_T_DOUBLE) return f64_of(x.v) == f64_of(y.v);
      if (x.t == CBH_T_LIST || x.t == CBH_T_MAP) { unsupported = true; return 4; }
      return x.v == y.v;
    }
    if (is_num(x.t) && is_num(y.t)) return num_cmp(x, y) == 0;
    return 0;
  }
  // one comparison: 0 / 1, 3 = CEL error, 4 = unsupported
  int compare(u32 op, V x, V y) {
    if (x.t == CBH_T_ERR || y.t == CBH_T_ERR) return 3;
    if (op == OP_EQ || op == OP_NE) {
      const int e = equal(x, y);
      return e == 4 ? 4 : (op == OP_EQ ? e : 1 - e);
    }
    if (op == OP_IN) {
      const u32 sel = (u32)(y.v >> 62), off = (u32)((y.v >> 32) & 0x3FFFFFFFu), n = (u32)y.v;
      if (y.t == CBH_T_LIST) {
        int found = 0;
        for (u32 i = 0; i < n; ++i) { const int e = equal(x, heap(sel, off + i)); if (e == 4) return 4; found |= e; }
        return found;
      }
      if (y.t == CBH_T_MAP) {
        int found = 0;
        for (u32 i = 0; i < n; ++i) { const int e = equal(x, heap(sel, off + 2 * i)); if (e == 4) return 4; found |= e; }
        return found;
      }
      return 3;   // no such overload
    }
    int c;
    if (is_num(x.t) && is_num(y.t)) { c = num_cmp(x, y); if (c == 2) return 0; }
    else if (x.t != y.t) return 3;
    else if (x.t == CBH_T_STRING) {
      const Str p = str_of(g, b, (u32)x.v), q = str_of(g, b, (u32)y.v);
      const int m = memcmp(p.p, q.p, std::min(p.n, q.n));
      c = m ? (m < 0 ? -1 : 1) : (p.n < q.n ? -1 : p.n > q.n ? 1 : 0);
    } else if (x.t == CBH_T_TIMESTAMP || x.t == CBH_T_DURATION) c = (i64)x.v < (i64)y.v ? -1 : (i64)x.v > (i64)y.v ? 1 : 0;
    else if (x.t == CBH_T_BOOL) c = (int)x.v - (int)y.v;
    else return 3;
    return op == OP_LT ? c < 0 : op == OP_LE ? c <= 0 : op == OP_GT ? c > 0 : c >= 0;
  }
  // EvalContext.SatisfiesCondition (check.go:679-756) for the fused-leaf forms.
  // Returns 0 / 1, 2 = strict-mode evaluation error; sets `err` when a CEL error was absorbed.
  int satisfies(u32 ref, bool strict, bool& err) {
    if (ref == CBH_NONE) return 1;
    if (!(ref & (CBH_COND_LEAF | CBH_COND_LEAFTREE))) { unsupported = true; return 0; }
    u32 pc = ref & CBH_COND_PC_MASK;
    bool live = true, last = false;
    u32 saved = 0, acc = 0, depth = 0;
    for (;;) {
      const u32 w = g.code[pc++], op = w & 0xFF, a = w >> 8;
      if (op == OP_LEAF_BIN) {
        const u32 a0 = g.code[pc], a1 = g.code[pc + 1]; pc += 2;
        last = false;
        if (live) {   // leaves after the deciding one are not evaluated
          const int v = compare(a & 0xFF, operand((a >> 8) & 0xF, a0), operand((a >> 12) & 0xF, a1));
          if (v == 4) { unsupported = true; return 0; }
          if (v == 3) { err = true; if (strict) return 2; }
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
          if (a == 0) { if (!last) { acc &= ~bit; live = false; } }      // all: first false decides
          else if (last) { acc |= bit; live = false; }                     // any / none: first true decides
        }
      } else if (op == OP_TREE_END) {
        --depth;
        const u32 bit = 1u << depth;
        live = (saved & bit) != 0;
        last = ((acc & bit) != 0) != (a == 2);
      } else break;   // OP_RET
    }
    return last ? 1 : 0;
  }
};

u32 chain_next(const Img& g, u32 si, u32 flag) {
  while (si != CBH_NONE && !(g.scope_flags[si] & flag)) si = g.scope_parent[si];
  return si;
}
// GetAllScopes (ruletable.go:848-882): the scopes of the chain that carry policies of this kind
std::vector<u32> scope_chain(const Img& g, u32 word, u32 flag, bool lenient) {
  std::vector<u32> out;
  const u32 si0 = word & ~CBH_SCOPE_EXACT;
  const bool exact = (word & CBH_SCOPE_EXACT) != 0;
  if (!lenient && !(exact && (g.scope_flags[si0] & flag))) return out;
  for (u32 si = chain_next(g, si0, flag); si != CBH_NONE; si = chain_next(g, g.scope_parent[si], flag)) out.push_back(si);
  return out;
}

struct CondMemo { u32 row; int r; bool err; };

void check_request(const Img& g, const cbh_batch& b, const cbh_params& p, u32 r, cbh_result& out) {
  const u32 NR = b.n_requests;
  auto RQ = [&](u32 f) { return b.req_u32[(size_t)f * NR + r]; };
  const u32 pid = RQ(CBH_RQ_PRINCIPAL_ID), kind = RQ(CBH_RQ_KIND), p_ver = RQ(CBH_RQ_P_VERSION), r_ver = RQ(CBH_RQ_R_VERSION);
  const u32 role_off = RQ(CBH_RQ_ROLE_OFF), role_cnt = RQ(CBH_RQ_ROLE_CNT), act_off = RQ(CBH_RQ_ACT_OFF), act_cnt = RQ(CBH_RQ_ACT_CNT);
  const bool lenient = p.flags & CBH_F_LENIENT_SCOPE_SEARCH, strict = p.flags & CBH_F_STRICT_EVALUATION;
  Req rq{g, b, r};

  const std::vector<u32> p_scopes = scope_chain(g, RQ(CBH_RQ_P_SCOPE), 2u, lenient);   // check.go:116-121
  const std::vector<u32> r_scopes = scope_chain(g, RQ(CBH_RQ_R_SCOPE), 1u, lenient);   // check.go:165-170
  bool p_exists = false, r_exists = false;
  for (u32 si : p_scopes) if (dir_find(g, CBH_B_PPEXISTS, p_ver, si, 0)) { p_exists = true; break; }
  const u64 kind_bits = glob_bits(g, b, 2, kind);
  for (u32 si : r_scopes) {                                                    // index.go:966-997: role-policy rows count too
    if (dir_find(g, CBH_B_RESEXISTS, r_ver, kind, si)) { r_exists = true; break; }
    if (const CbhHashSlot* rp = dir_find(g, CBH_B_RPRES, r_ver, si, 0))
      for (u32 k = 0; k < rp->v1 && !r_exists; ++k) r_exists = pat_match(g.pool[rp->v0 + k], kind, kind_bits);
    if (r_exists) break;
  }
  // AddParentRoles (index.go:716-742) for the request's effective resource scope: the ancestors of a role
  const u32 r_scope_w = RQ(CBH_RQ_R_SCOPE);
  const bool has_parents = (g.meta[CBH_M_FLAGS] & CBH_MF_HAS_PARENT_ROLES) && (r_scope_w & CBH_SCOPE_EXACT);
  auto ancestors = [&](u32 role, u32& off, u32& cnt) {
    off = cnt = 0;
    if (!has_parents) return;
    if (const CbhHashSlot* pv = dir_find(g, CBH_B_PARENTS, r_scope_w & ~CBH_SCOPE_EXACT, role, 0)) { off = pv->v0; cnt = pv->v1; }
  };
  const bool nothing = (p_scopes.empty() && r_scopes.empty()) || (!p_exists && !r_exists);

  std::vector<CondMemo> memo;            // conditionCache of the request (check.go:186, 316-340)
  std::vector<std::pair<u32, u64>> sdr;  // processedScopedDerivedRoles: scope -> effective derived roles (check.go:237-282)
  std::vector<std::pair<u32, bool>> sdr_err;
  u64 edr_acc = 0;
  auto cond_pair = [&](u32 row, u32 drcond, u32 cond, bool& err) -> int {
    for (const CondMemo& m : memo) if (m.row == row) { err = err || m.err; return m.r; }
    bool e = false;
    int res = rq.satisfies(drcond, strict, e);
    if (res == 1) res = rq.satisfies(cond, strict, e);
    memo.push_back(CondMemo{row, res, e});
    err = err || e;
    return res;
  };

  for (u32 k = 0; k < act_cnt; ++k) {
    const u32 t = act_off + k, action = b.tuple_action[t];
    const u64 act_bits = glob_bits(g, b, 0, action);
    u32 eff = 0, pol = nothing ? ((u32)CBH_P_NO_MATCH << 28) : ((u32)CBH_P_EMPTY << 28), scp = CBH_NONE;   // 0 = NO_MATCH
    bool err = false, done = nothing;
    for (u32 pt = 0; pt < 2 && !done; ++pt) {                                   // check.go:195
      const bool is_res = pt == 1;
      const std::vector<u32>& scopes = is_res ? r_scopes : p_scopes;
      const bool exists = is_res ? r_exists : p_exists;
      const u32 main_key = scopes.empty() ? 0u : (((u32)(is_res ? CBH_P_RESOURCE : CBH_P_PRINCIPAL) << 28) | scopes[0]);
      eff = 0;                                                                   // :206, policy / scope are NOT reset
      for (u32 ri = 0; ri < role_cnt; ++ri) {                                    // :208
        if (ri > 0 && !is_res) break;
        const u32 role = b.roles[role_off + ri];
        const u64 role_bits = glob_bits(g, b, 1, role);
        u32 anc_off = 0, anc_cnt = 0;
        if (is_res) ancestors(role, anc_off, anc_cnt);
        auto role_match = [&](u32 pat) {   // the pattern against [role] ++ its ancestors (check.go:293, index.go:214-336)
          if (pat_match(pat, role, role_bits)) return true;
          for (u32 q = 0; q < anc_cnt; ++q) if (pat_match(pat, g.pool[anc_off + q], glob_bits(g, b, 1, g.pool[anc_off + q]))) return true;
          return false;
        };
        bool has_allow = false;
        u32 r_eff = 0, r_pol = exists && !scopes.empty() ? main_key : ((u32)CBH_P_NO_MATCH << 28), r_scp = CBH_NONE;
        for (u32 si : scopes) {                                                  // :231
          u64 edr = 0; bool edr_err = false;
          if (is_res) {                                                          // :237-282, once per scope and request
            bool have = false;
            for (size_t i = 0; i < sdr.size(); ++i) if (sdr[i].first == si) { edr = sdr[i].second; edr_err = sdr_err[i].second; have = true; }
            if (!have) {
              if (const CbhHashSlot* bk = dir_find(g, CBH_B_RESOURCE, r_ver, kind, si)) {
                for (u32 d = bk->v2; d < bk->v2 + bk->v3; ++d) {
                  const u32* dr = g.dr + 4 * (size_t)d;
                  bool applies = dr[CBH_DR_PARENTS_CNT] == CBH_NONE;
                  for (u32 q = 0; !applies && q < dr[CBH_DR_PARENTS_CNT]; ++q)
                    for (u32 x = 0; !applies && x < role_cnt; ++x) {              // includingParentRoles (check.go:244)
                      const u32 want = g.pool[dr[CBH_DR_PARENTS_OFF] + q], have = b.roles[role_off + x];
                      applies = want == have;
                      u32 ao, ac; ancestors(have, ao, ac);
                      for (u32 z = 0; !applies && z < ac; ++z) applies = g.pool[ao + z] == want;
                    }
                  if (!applies) continue;
                  bool e = false;
                  const int res = rq.satisfies(dr[CBH_DR_COND], strict, e);
                  if (e) err = true;   // recorded in evaluation_errors like any condition's (check.go:262-276)
                  if (res == 2) edr_err = true; else if (res == 1) edr |= 1ull << dr[CBH_DR_NAME];
                }
              }
              sdr.push_back({si, edr}); sdr_err.push_back({si, edr_err});
              edr_acc |= edr;
            }
          }
          if (r_eff != 0) break;                                                 // :284
          bool brk = false;
          // baseBM (index.go:250-305): some binding at (version, scope) - a resource-policy row or a role-policy
          // row - matches the resource and one of [role] ++ ancestors; otherwise Query returns before
          // appendRolePolicyDenies and the scope contributes nothing.
          bool base = false;
          if (is_res && (g.meta[CBH_M_FLAGS] & CBH_MF_HAS_ROLE_POLICIES)) {
            if (const CbhHashSlot* bk0 = dir_find(g, CBH_B_RESOURCE, r_ver, kind, si))
              for (u32 row = bk0->v0; row < bk0->v0 + bk0->v1 && !base; ++row) {
                const u32* rw = g.rows + CBH_ROW_NF * (size_t)row;
                const u32* pt = g.rowpat + CBH_PAT_NF * (size_t)row;
                const u32 n_role = pt[CBH_PAT_COUNTS] >> 16;
                if (n_role == 0) base = role_match(pt[CBH_PAT_ROLE]);
                else for (u32 i = 0; i < n_role && !base; ++i)
                  base = role_match((rw[CBH_ROW_FLAGS] & CBH_ROW_F_ROLE_LIST) ? g.pool[pt[CBH_PAT_ROLE] + i] : (i == 0 ? pt[CBH_PAT_ROLE] : pt[CBH_PAT_R1 + i - 1]));
              }
            for (u32 k = 0; k <= anc_cnt && !base; ++k) {
              const CbhHashSlot* rp = dir_find(g, CBH_B_ROLEPOL, r_ver, si, k == 0 ? role : g.pool[anc_off + k - 1]);
              for (u32 row = rp ? rp->v0 : 0; rp && row < rp->v0 + rp->v1 && !base; ++row)
                base = pat_match(g.rprows[CBH_RP_NF * (size_t)row + CBH_RP_RESOURCE], kind, kind_bits);
            }
          }
          if (base) {
            // synthetic DENY bindings from the role policies of [role] ++ ancestors come first (index.go:352-530)
            for (u32 k = 0; k <= anc_cnt && !brk && !done; ++k) {
              const u32 srole = k == 0 ? role : g.pool[anc_off + k - 1];
              const CbhHashSlot* rp = dir_find(g, CBH_B_ROLEPOL, r_ver, si, srole);
              if (!rp) continue;
              const u32 rp_pol = ((u32)CBH_P_TABLE << 28) | rp->v2;
              auto rule_matches = [&](const u32* rr) {   // the rule is for this resource and allows this action
                if (!pat_match(rr[CBH_RP_RESOURCE], kind, kind_bits)) return false;
                for (u32 a = 0; a < (rr[CBH_RP_ALLOW_CNT] & CBH_RP_CNT_MASK); ++a) if (pat_match(g.pool[rr[CBH_RP_ALLOW_OFF] + a], action, act_bits)) return true;
                return false;
              };
              bool any = false;
              for (u32 row = rp->v0; row < rp->v0 + rp->v1 && !any; ++row) any = rule_matches(g.rprows + CBH_RP_NF * (size_t)row);
              bool deny = !any;                          // no rule for the resource / no allow-action matched (index.go:436-461)
              for (u32 row = rp->v0; row < rp->v0 + rp->v1 && !deny; ++row) {
                const u32* rr = g.rprows + CBH_RP_NF * (size_t)row;
                if (rr[CBH_RP_COND] == CBH_NONE || !rule_matches(rr)) continue;
                bool e = false;
                const int res = rq.satisfies(rr[CBH_RP_COND], strict, e);   // the binding is DENY when none(condition)
                if (e) err = true;
                if (res == 2) { eff = CBH_EFFECT_DENY; pol = rp_pol; scp = si; done = true; break; }
                if (res == 0) deny = true;
              }
              if (done) break;
              if (deny) { r_eff = CBH_EFFECT_DENY; r_pol = rp_pol; r_scp = si; brk = true; }   // check.go:395-403
            }
          }
          if (done || brk) break;
          const CbhHashSlot* bk = is_res ? dir_find(g, CBH_B_RESOURCE, r_ver, kind, si)
                                         : dir_find(g, CBH_B_PRINCIPAL, r_ver, si, pid);   // resource version: check.go:294
          for (u32 row = bk ? bk->v0 : 0; bk && row < bk->v0 + bk->v1; ++row) {  // :295-414, binding order
            const u32* rw = g.rows + CBH_ROW_NF * (size_t)row;
            const u32* pt = g.rowpat + CBH_PAT_NF * (size_t)row;   // the pattern references: this restatement never uses the class masks
            const u32 fl = rw[CBH_ROW_FLAGS];
            const u32 n_act = pt[CBH_PAT_COUNTS] & 0xFFFFu, n_role = pt[CBH_PAT_COUNTS] >> 16;   // 0 = one reference
            // the i-th pattern of a list: in the pool, or inline in the record (first word + CBH_ROW_A1 / R1 ...)
            auto nth = [&](u32 first, u32 more, bool in_pool, u32 i) { return in_pool ? g.pool[pt[first] + i] : (i == 0 ? pt[first] : pt[more + i - 1]); };
            bool m;
            if (!is_res) m = pat_match(pt[CBH_PAT_RESOURCE], kind, kind_bits);
            else if (n_role == 0) m = role_match(pt[CBH_PAT_ROLE]);
            else { m = false; for (u32 i = 0; i < n_role; ++i) m = m || role_match(nth(CBH_PAT_ROLE, CBH_PAT_R1, fl & CBH_ROW_F_ROLE_LIST, i)); }
            if (!m) continue;
            if (n_act == 0) m = pat_match(pt[CBH_PAT_ACTION], action, act_bits);
            else { m = false; for (u32 i = 0; i < n_act; ++i) m = m || pat_match(nth(CBH_PAT_ACTION, CBH_PAT_A1, fl & CBH_ROW_F_ACTION_LIST, i), action, act_bits); }
            if (!m) continue;
            const u32 e = fl & 3u;
            const int res = cond_pair(row, rw[CBH_ROW_DRCOND], rw[CBH_ROW_COND], err);
            if (res == 2) {                                                      // :353-356, 371-374
              eff = CBH_EFFECT_DENY; pol = ((u32)CBH_P_TABLE << 28) | rw[CBH_ROW_POLICY]; scp = si; done = true; brk = true;
              break;
            }
            if (res != 1) continue;
            if (e == CBH_EFFECT_ALLOW) has_allow = true;
            else if (e == CBH_EFFECT_DENY) { r_eff = CBH_EFFECT_DENY; r_scp = si; brk = true; break; }   // :392-403
          }
          if (done || brk) break;
          if (has_allow) {                                                       // :416-425
            const u32 sp = (g.scope_flags[si] >> 2) & 3u;
            if (sp == 2u) has_allow = false;                                     // REQUIRE_PARENTAL_CONSENT_FOR_ALLOWS
            else if (sp == 1u) { r_eff = CBH_EFFECT_ALLOW; r_scp = si; break; }  // OVERRIDE_PARENT
          }
        }
        if (done) break;
        if (eff == 0) { eff = r_eff; pol = r_pol; scp = r_scp; }                 // :429-431
        if (r_eff == CBH_EFFECT_ALLOW) { eff = r_eff; pol = r_pol; scp = r_scp; break; }   // :433-436
      }
      if (eff != 0) break;                                                       // :445-448
    }
    out.effect[t] = (u8)(eff == CBH_EFFECT_ALLOW ? CBH_EFFECT_ALLOW : CBH_EFFECT_DENY);   // NO_MATCH -> DENY (:451-453)
    if (out.policy) out.policy[t] = pol;
    if (out.scope) out.scope[t] = scp;
    if (out.status) out.status[t] = err ? CBH_ST_CEL_ERROR : CBH_ST_OK;
  }
  if (rq.unsupported && out.status) for (u32 k = 0; k < act_cnt; ++k) out.status[act_off + k] = CBH_ST_UNSUPPORTED;
  if (out.edr_mask) out.edr_mask[r] = edr_acc;
}

}  // namespace

// 0 = done, -1 = bad image (1 is reserved for "table outside this restatement"; no such table today)
extern "C" int ccheck_run(const void* blob, size_t len, const cbh_batch* in, const cbh_params* p, cbh_result* out, int n_threads) {
  Img g;
  if (!parse(g, (const u8*)blob, len)) return -1;
  const u32 n = in->n_requests;
  if (n_threads <= 1) {
    for (u32 r = 0; r < n; ++r) check_request(g, *in, *p, r, *out);
    return 0;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&, t]() {   // contiguous request ranges; requests are independent (engine.go:296-304)
      const u32 lo = (u32)((u64)n * t / n_threads), hi = (u32)((u64)n * (t + 1) / n_threads);
      for (u32 r = lo; r < hi; ++r) check_request(g, *in, *p, r, *out);
    });
  for (auto& x : th) x.join();
  return 0;
}
